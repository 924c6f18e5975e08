// HBM-bound helper kernels of the dl4ds train step (gfx950): ReLU-mask + bias-gradient reduction,
// strided/d2s view copies (concat, residual gradients), activations, depth_to_space, 2x2 max-pool,
// bilinear resize, per-pixel locally-connected 1x1 layer.  All fp32 NHWC.
//
// Reference call sites (third-party TF ops there): blocks.py:75 (Activation), :228 (Add), :276/:656
// (Concatenate), :427 (depth_to_space), :489 (Resizing), :613 (MaxPooling2D), :322-328 (LocallyConnected2D).
#include "ops.h"
#include "prof.h"
#include <algorithm>

namespace {

__device__ __forceinline__ void unflatten_pix(const TView& v, size_t pix, int& n, int& y, int& x) {
    x = (int)(pix % v.W);
    size_t r = pix / v.W;
    y = (int)(r % v.H);
    n = (int)(r / v.H);
}

// ------------------------------------------------------------------------------------------
// dz = dy * [y > 0]; partial[blockIdx.x][c] = sum over this block's pixels of dz[..., c]
template <int TX>
__global__ void __launch_bounds__(256) bias_act_bwd_kernel(TView dy, TView y, TView dz, float* partial,
                                                           size_t npix) {
    constexpr int TY = 256 / TX;
    __shared__ float red[TY][TX + 1];
    const int tx = threadIdx.x % TX, tyi = threadIdx.x / TX;
    const int c = blockIdx.y * TX + tx;
    float sum = 0.f;
    if (c < dy.C) {
        for (size_t pix = (size_t)blockIdx.x * TY + tyi; pix < npix; pix += (size_t)gridDim.x * TY) {
            int n, yy, xx;
            unflatten_pix(dy, pix, n, yy, xx);
            float v = dy.p[view_off(dy, n, yy, xx, c)];
            if (y.p) v = (y.p[view_off(y, n, yy, xx, c)] > 0.f) ? v : 0.f;
            if (dz.p) dz.p[view_off(dz, n, yy, xx, c)] = v;
            sum += v;
        }
    }
    red[tyi][tx] = sum;
    __syncthreads();
    if (tyi == 0 && partial) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < TY; ++k) s += red[k][tx];
        if (c < dy.C) partial[(size_t)blockIdx.x * dy.C + c] = s;
    }
}

__global__ void reduce_slabs_kernel2(const float* __restrict__ partial, float* __restrict__ out, size_t n,
                                     int S, int accumulate) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < S; ++k) s += partial[(size_t)k * n + e];
        out[e] = accumulate ? out[e] + s : s;
    }
}

// mask-only fast path on contiguous tensors: dz = dy * [y > 0], 16 bytes per lane
__global__ void __launch_bounds__(256) relu_mask_flat4_kernel(const float4* __restrict__ dy, const float4* __restrict__ y,
                                                              float4* __restrict__ dz, size_t n4) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (size_t)gridDim.x * blockDim.x) {
        float4 g = dy[e];
        const float4 m = y[e];
        g.x = m.x > 0.f ? g.x : 0.f;
        g.y = m.y > 0.f ? g.y : 0.f;
        g.z = m.z > 0.f ? g.z : 0.f;
        g.w = m.w > 0.f ? g.w : 0.f;
        dz[e] = g;
    }
}

// dst (+)= dy * [y > 0] on contiguous tensors: the ReLU backward of an Add operand folded into the Add's gradient copy
__global__ void __launch_bounds__(256) masked_axpy4_kernel(const float4* __restrict__ dy, const float4* __restrict__ y,
                                                           float4* __restrict__ dst, size_t n4, int accumulate) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (size_t)gridDim.x * blockDim.x) {
        float4 g = dy[e];
        const float4 m = y[e];
        g.x = m.x > 0.f ? g.x : 0.f;
        g.y = m.y > 0.f ? g.y : 0.f;
        g.z = m.z > 0.f ? g.z : 0.f;
        g.w = m.w > 0.f ? g.w : 0.f;
        if (accumulate) {
            const float4 o = dst[e];
            g.x += o.x; g.y += o.y; g.z += o.z; g.w += o.w;
        }
        dst[e] = g;
    }
}
// both operands of an Add in one pass: da (+)= dy * [ya > 0], db (+)= dy * [yb > 0]  (dy is read once: five tensor passes instead of six)
__global__ void __launch_bounds__(256) masked_axpy4_pair_kernel(const float4* __restrict__ dy, const float4* __restrict__ ya, float4* __restrict__ da,
                                                                const float4* __restrict__ yb, float4* __restrict__ db, size_t n4, int acc_a, int acc_b) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (size_t)gridDim.x * blockDim.x) {
        const float4 g = dy[e], ma = ya[e], mb = yb[e];
        float4 ga, gb;
        ga.x = ma.x > 0.f ? g.x : 0.f; ga.y = ma.y > 0.f ? g.y : 0.f; ga.z = ma.z > 0.f ? g.z : 0.f; ga.w = ma.w > 0.f ? g.w : 0.f;
        gb.x = mb.x > 0.f ? g.x : 0.f; gb.y = mb.y > 0.f ? g.y : 0.f; gb.z = mb.z > 0.f ? g.z : 0.f; gb.w = mb.w > 0.f ? g.w : 0.f;
        if (acc_a) { const float4 o = da[e]; ga.x += o.x; ga.y += o.y; ga.z += o.z; ga.w += o.w; }
        if (acc_b) { const float4 o = db[e]; gb.x += o.x; gb.y += o.y; gb.z += o.z; gb.w += o.w; }
        da[e] = ga;
        db[e] = gb;
    }
}
__global__ void masked_axpy1_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dst, size_t n,
                                    int accumulate) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const float g = y[e] > 0.f ? dy[e] : 0.f;
        dst[e] = accumulate ? dst[e] + g : g;
    }
}

int bias_blocks(size_t npix, int TY) { return (int)std::min<size_t>(cdivz(npix, (size_t)TY * 8), 1024); }

int pick_tx(int C) { return C <= 8 ? 8 : (C <= 16 ? 16 : (C <= 32 ? 32 : 64)); }

// ------------------------------------------------------------------------------------------
__global__ void view_axpy_kernel(TView src, TView dst, float alpha, int accumulate, size_t total) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % src.C);
        int n, y, x;
        unflatten_pix(src, e / src.C, n, y, x);
        const float v = alpha * src.p[view_off(src, n, y, x, c)];
        const size_t o = view_off(dst, n, y, x, c);
        dst.p[o] = accumulate ? dst.p[o] + v : v;
    }
}

// channel-slice copies (Concatenate forward / backward: one side is a [C]-wide slice of a wider pixel): float4 per thread,
// pixel-major so that a wave covers whole pixels back to back; pixel strides in floats, images contiguous (nstride = H*W*ld)
__global__ void strided_axpy4_kernel(const float* __restrict__ src, int ld_s, float* __restrict__ dst, int ld_d, int c4n,
                                     size_t step_pix, int step_c4, float alpha, int accumulate, size_t total4) {
    // element e = pix * c4n + c4; one division per thread, then (pix, c4) advance by the grid stride's quotient / remainder
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t pix = e / (size_t)c4n;
    int c4 = (int)(e - pix * (size_t)c4n);
    for (; e < total4; e += (size_t)gridDim.x * blockDim.x) {
        float4 v = *reinterpret_cast<const float4*>(src + pix * (size_t)ld_s + (size_t)c4 * 4);
        v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha;
        float4* d = reinterpret_cast<float4*>(dst + pix * (size_t)ld_d + (size_t)c4 * 4);
        if (accumulate) {
            const float4 o = *d;
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        *d = v;
        pix += step_pix; c4 += step_c4;
        if (c4 >= c4n) { c4 -= c4n; ++pix; }
    }
}

// the same for slices whose channel counts / pixel pitches are not multiples of four (a 26-channel concatenation of 16 + 8 + 2):
// V = 2 (everything even) or 1 floats per thread; no per-element divisions (the generic view kernel spends ~40 instructions per
// element on them)
template <int V>
__global__ void strided_axpy_small_kernel(const float* __restrict__ src, int ld_s, float* __restrict__ dst, int ld_d, int cvn,
                                          size_t step_pix, int step_cv, float alpha, int accumulate, size_t totalv) {
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t pix = e / (size_t)cvn;
    int cv = (int)(e - pix * (size_t)cvn);
    for (; e < totalv; e += (size_t)gridDim.x * blockDim.x) {
        const float* sp = src + pix * (size_t)ld_s + (size_t)cv * V;
        float* dp = dst + pix * (size_t)ld_d + (size_t)cv * V;
        if constexpr (V == 2) {
            float2 v = *reinterpret_cast<const float2*>(sp);
            v.x *= alpha; v.y *= alpha;
            if (accumulate) { const float2 o = *reinterpret_cast<const float2*>(dp); v.x += o.x; v.y += o.y; }
            *reinterpret_cast<float2*>(dp) = v;
        } else {
            float v = alpha * sp[0];
            if (accumulate) v += dp[0];
            dp[0] = v;
        }
        pix += step_pix; cv += step_cv;
        if (cv >= cvn) { cv -= cvn; ++pix; }
    }
}

// dst (+)= src * [mask > 0]: float4, strided pixels on all three sides (see strided_axpy4_kernel)
__global__ void strided_masked_axpy4_kernel(const float* __restrict__ src, int ld_s, const float* __restrict__ mask, int ld_m,
                                            float* __restrict__ dst, int ld_d, int c4n, size_t step_pix, int step_c4,
                                            int accumulate, size_t total4) {
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t pix = e / (size_t)c4n;
    int c4 = (int)(e - pix * (size_t)c4n);
    for (; e < total4; e += (size_t)gridDim.x * blockDim.x) {
        float4 v = *reinterpret_cast<const float4*>(src + pix * (size_t)ld_s + (size_t)c4 * 4);
        const float4 m = *reinterpret_cast<const float4*>(mask + pix * (size_t)ld_m + (size_t)c4 * 4);
        v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
        float4* d = reinterpret_cast<float4*>(dst + pix * (size_t)ld_d + (size_t)c4 * 4);
        if (accumulate) {
            const float4 o = *d;
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        *d = v;
        pix += step_pix; c4 += step_c4;
        if (c4 >= c4n) { c4 -= c4n; ++pix; }
    }
}
__global__ void view_masked_axpy_kernel(TView src, TView mask, TView dst, int accumulate, size_t total) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % src.C);
        int n, y, x;
        unflatten_pix(src, e / src.C, n, y, x);
        float v = src.p[view_off(src, n, y, x, c)];
        v = mask.p[view_off(mask, n, y, x, c)] > 0.f ? v : 0.f;
        const size_t o = view_off(dst, n, y, x, c);
        dst.p[o] = accumulate ? dst.p[o] + v : v;
    }
}

__global__ void flat_axpy4_kernel(const float4* __restrict__ src, float4* __restrict__ dst, float alpha,
                                  int accumulate, size_t n4) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (size_t)gridDim.x * blockDim.x) {
        float4 v = src[e];
        v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha;
        if (accumulate) {
            float4 d = dst[e];
            v.x += d.x; v.y += d.y; v.z += d.z; v.w += d.w;
        }
        dst[e] = v;
    }
}

__global__ void add_act_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                               size_t n, int relu) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        float v = a[e] + b[e];
        out[e] = relu ? fmaxf(v, 0.f) : v;
    }
}

__device__ __forceinline__ float act_f(float x, int kind) {
    switch (kind) {
        case ACT_RELU: return fmaxf(x, 0.f);
        case ACT_SIGMOID: return 1.f / (1.f + expf(-x));
        case ACT_TANH: return tanhf(x);
        case ACT_ELU: return x > 0.f ? x : expm1f(x);
        case ACT_LEAKY_RELU: return x > 0.f ? x : 0.2f * x;
        case ACT_SELU: return 1.0507009873554805f * (x > 0.f ? x : 1.6732632423543772f * expm1f(x));
        case ACT_GELU: return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
        default: return x;
    }
}
__device__ __forceinline__ float act_df(float x, int kind) {
    switch (kind) {
        case ACT_RELU: return x > 0.f ? 1.f : 0.f;
        case ACT_SIGMOID: { float s = 1.f / (1.f + expf(-x)); return s * (1.f - s); }
        case ACT_TANH: { float t = tanhf(x); return 1.f - t * t; }
        case ACT_ELU: return x > 0.f ? 1.f : expf(x);
        case ACT_LEAKY_RELU: return x > 0.f ? 1.f : 0.2f;
        case ACT_SELU: return 1.0507009873554805f * (x > 0.f ? 1.f : 1.6732632423543772f * expf(x));
        case ACT_GELU: {
            const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
            const float pdf = 0.3989422804014327f * expf(-0.5f * x * x);
            return cdf + x * pdf;
        }
        default: return 1.f;
    }
}
__global__ void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, int kind) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x)
        y[e] = act_f(x[e], kind);
}
__global__ void act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                               size_t n, int kind, int accumulate) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const float g = dy[e] * act_df(x[e], kind);
        dx[e] = accumulate ? dx[e] + g : g;
    }
}
__global__ void fill_kernel(float* p, size_t n, float v) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) p[e] = v;
}

// ------------------------------------------------------------------------------------------
__global__ void maxpool2_fwd_kernel(TView x, TView y, size_t total) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % y.C);
        int n, oy, ox;
        unflatten_pix(y, e / y.C, n, oy, ox);
        float m = x.p[view_off(x, n, 2 * oy, 2 * ox, c)];
        m = fmaxf(m, x.p[view_off(x, n, 2 * oy, 2 * ox + 1, c)]);
        m = fmaxf(m, x.p[view_off(x, n, 2 * oy + 1, 2 * ox, c)]);
        m = fmaxf(m, x.p[view_off(x, n, 2 * oy + 1, 2 * ox + 1, c)]);
        y.p[view_off(y, n, oy, ox, c)] = m;
    }
}
// one thread per OUTPUT element writes its 2x2 input window (windows are disjoint -> no atomics);
// gradient goes to the first maximum in row-major window order.  Rows/cols dropped by the VALID
// pooling (odd H or W) must have been zero-filled / left untouched by the caller.
__global__ void maxpool2_bwd_kernel(TView x, TView y, TView dy, TView dx, int accumulate, int relu_mask, size_t total) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % y.C);
        int n, oy, ox;
        unflatten_pix(y, e / y.C, n, oy, ox);
        const float m = y.p[view_off(y, n, oy, ox, c)];
        const float g = dy.p[view_off(dy, n, oy, ox, c)];
        bool found = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int iy = 2 * oy + (k >> 1), ix = 2 * ox + (k & 1);
            const float xv = x.p[view_off(x, n, iy, ix, c)];
            float gv = 0.f;
            if (!found && xv == m) { gv = g; found = true; }
            if (relu_mask && !(xv > 0.f)) gv = 0.f;
            const size_t o = view_off(dx, n, iy, ix, c);
            dx.p[o] = accumulate ? dx.p[o] + gv : gv;
        }
    }
}

// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void bilinear_src(int o, float scale, int in_size, int& lo, int& hi, float& f) {
    const float src = ((float)o + 0.5f) * scale - 0.5f;
    const float fl = floorf(src);
    lo = max((int)fl, 0);
    hi = min((int)ceilf(src), in_size - 1);
    f = src - fl;
}
__global__ void resize_fwd_kernel(TView x, TView y, float sy, float sx, size_t total) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % y.C);
        int n, oy, ox;
        unflatten_pix(y, e / y.C, n, oy, ox);
        int y0, y1, x0, x1;
        float fy, fx;
        bilinear_src(oy, sy, x.H, y0, y1, fy);
        bilinear_src(ox, sx, x.W, x0, x1, fx);
        const float v00 = x.p[view_off(x, n, y0, x0, c)], v01 = x.p[view_off(x, n, y0, x1, c)];
        const float v10 = x.p[view_off(x, n, y1, x0, c)], v11 = x.p[view_off(x, n, y1, x1, c)];
        const float top = v00 * (1.f - fx) + v01 * fx;
        const float bot = v10 * (1.f - fx) + v11 * fx;
        y.p[view_off(y, n, oy, ox, c)] = top * (1.f - fy) + bot * fy;
    }
}
// Resizing(..., interpolation='nearest') = tf.image.resize(method='nearest') with half-pixel centres:
// src = min(floor((dst + 0.5) * in / out), in - 1)   (blocks.py:473-489 with rc_interpolation='nearest')
__device__ __forceinline__ int nearest_src(int o, float scale, int in_size) {
    return min((int)floorf(((float)o + 0.5f) * scale), in_size - 1);
}
__global__ void resize_nearest_fwd_kernel(TView x, TView y, float sy, float sx, size_t total) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % y.C);
        int n, oy, ox;
        unflatten_pix(y, e / y.C, n, oy, ox);
        y.p[view_off(y, n, oy, ox, c)] = x.p[view_off(x, n, nearest_src(oy, sy, x.H), nearest_src(ox, sx, x.W), c)];
    }
}
// gather form: one thread per INPUT element sums the outputs that copied it (those within one source step of it)
__global__ void resize_nearest_bwd_kernel(TView dy, TView dx, float sy, float sx, int accumulate, size_t total) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % dx.C);
        int n, iy, ix;
        unflatten_pix(dx, e / dx.C, n, iy, ix);
        const int oy_lo = max(0, (int)floorf((float)iy / sy - 0.5f) - 1), oy_hi = min(dy.H - 1, (int)ceilf((float)(iy + 1) / sy - 0.5f) + 1);
        const int ox_lo = max(0, (int)floorf((float)ix / sx - 0.5f) - 1), ox_hi = min(dy.W - 1, (int)ceilf((float)(ix + 1) / sx - 0.5f) + 1);
        float g = 0.f;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            if (nearest_src(oy, sy, dx.H) != iy) continue;
            for (int ox = ox_lo; ox <= ox_hi; ++ox)
                if (nearest_src(ox, sx, dx.W) == ix) g += dy.p[view_off(dy, n, oy, ox, c)];
        }
        const size_t o = view_off(dx, n, iy, ix, c);
        dx.p[o] = accumulate ? dx.p[o] + g : g;
    }
}
// gather form (deterministic): one thread per INPUT element sums the output pixels that read it
__global__ void resize_bwd_kernel(TView dy, TView dx, float sy, float sx, int accumulate, size_t total) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % dx.C);
        int n, iy, ix;
        unflatten_pix(dx, e / dx.C, n, iy, ix);
        // candidate outputs: src in (iy-1, iy+1)  <=>  o in ((iy-0.5)/s-0.5, (iy+1.5)/s-0.5)
        const int oy_lo = max(0, (int)floorf(((float)iy - 0.5f) / sy - 0.5f) - 1);
        const int oy_hi = min(dy.H - 1, (int)ceilf(((float)iy + 1.5f) / sy - 0.5f) + 1);
        const int ox_lo = max(0, (int)floorf(((float)ix - 0.5f) / sx - 0.5f) - 1);
        const int ox_hi = min(dy.W - 1, (int)ceilf(((float)ix + 1.5f) / sx - 0.5f) + 1);
        float g = 0.f;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            int y0, y1; float fy;
            bilinear_src(oy, sy, dx.H, y0, y1, fy);
            float wy = 0.f;
            if (y0 == iy) wy += 1.f - fy;
            if (y1 == iy) wy += fy;
            if (wy == 0.f) continue;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                int x0, x1; float fx;
                bilinear_src(ox, sx, dx.W, x0, x1, fx);
                float wx = 0.f;
                if (x0 == ix) wx += 1.f - fx;
                if (x1 == ix) wx += fx;
                if (wx == 0.f) continue;
                g += wy * wx * dy.p[view_off(dy, n, oy, ox, c)];
            }
        }
        const size_t o = view_off(dx, n, iy, ix, c);
        dx.p[o] = accumulate ? dx.p[o] + g : g;
    }
}

// ------------------------------------------------------------------------------------------
constexpr int kLcMax = 8;
__global__ void localconv_fwd_kernel(TView x, const float* __restrict__ w, const float* __restrict__ b, TView y,
                                     size_t npix) {
    const int C = x.C, F = y.C;
    for (size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (size_t)gridDim.x * blockDim.x) {
        int n, yy, xx;
        unflatten_pix(x, pix, n, yy, xx);
        const size_t hw = (size_t)yy * x.W + xx;
        float xv[kLcMax];
        for (int c = 0; c < C; ++c) xv[c] = x.p[view_off(x, n, yy, xx, c)];
        for (int f = 0; f < F; ++f) {
            float s = b ? b[hw * F + f] : 0.f;
            for (int c = 0; c < C; ++c) s += xv[c] * w[(hw * C + c) * F + f];
            y.p[view_off(y, n, yy, xx, f)] = s;
        }
    }
}
// one thread per (h,w): loops the batch, so dW / db need no atomics
// One thread per (grid point, quarter of the batch): the weight gradient of a grid point is a sum over the batch, and one thread
// walking all N images of its point (128 dependent load round trips, 4 waves per CU) was latency-bound at 0.6 TB/s.  A wave
// holds 16 points x 4 batch lanes; the four partial sums meet through two shuffles.  CT / FT > 0: compile-time channel counts
// (registers instead of dynamically indexed scratch arrays); 0: run-time counts up to kLcMax.
template <int CT, int FT>
__global__ void __launch_bounds__(256) localconv_bwd_kernel(TView x, const float* __restrict__ w, TView dy, TView dx, int acc_dx,
                                                            float* __restrict__ dw, float* __restrict__ db, int acc_dw) {
    constexpr int CM = CT ? CT : kLcMax, FM = FT ? FT : kLcMax;
    const int C = CT ? CT : x.C, F = FT ? FT : dy.C;
    const size_t nhw = (size_t)x.H * x.W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pl = lane & 15, nl = lane >> 4;
    const size_t hw = ((size_t)blockIdx.x * 4 + wave) * 16 + pl;
    const bool live = hw < nhw;
    const size_t hwc = live ? hw : 0;
    const int yy = (int)(hwc / x.W), xx = (int)(hwc % x.W);
    float wl[CM * FM], gw[CM * FM], gb[FM];
#pragma unroll
    for (int i = 0; i < CM * FM; ++i) { wl[i] = (i < C * F) ? w[hwc * C * F + i] : 0.f; gw[i] = 0.f; }
#pragma unroll
    for (int f = 0; f < FM; ++f) gb[f] = 0.f;
    for (int n = nl; n < x.N; n += 4) {
        float xv[CM], gy[FM];
#pragma unroll
        for (int c = 0; c < CM; ++c) xv[c] = (c < C) ? x.p[view_off(x, n, yy, xx, c)] : 0.f;
#pragma unroll
        for (int f = 0; f < FM; ++f) { gy[f] = (f < F) ? dy.p[view_off(dy, n, yy, xx, f)] : 0.f; gb[f] += gy[f]; }
#pragma unroll
        for (int c = 0; c < CM; ++c) {
            if (c < C) {
                float sdx = 0.f;
#pragma unroll
                for (int f = 0; f < FM; ++f)
                    if (f < F) { sdx += gy[f] * wl[c * F + f]; gw[c * F + f] += xv[c] * gy[f]; }
                if (dx.p && live) {
                    const size_t o = view_off(dx, n, yy, xx, c);
                    dx.p[o] = acc_dx ? dx.p[o] + sdx : sdx;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < CM * FM; ++i) {
        float v = gw[i];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (nl == 0 && live && i < C * F) dw[hw * C * F + i] = acc_dw ? dw[hw * C * F + i] + v : v;
    }
    if (db) {
#pragma unroll
        for (int f = 0; f < FM; ++f) {
            float v = gb[f];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (nl == 0 && live && f < F) db[hw * F + f] = acc_dw ? db[hw * F + f] + v : v;
        }
    }
}

inline int ew_blocks(size_t n) { return (int)std::max<size_t>(1, std::min<size_t>(cdivz(n, 256), 8192)); }
inline bool plain_contig(const TView& v) { return v.d2s <= 1 && v.ld == v.C && v.nstride == (size_t)v.H * v.W * v.C; }

}  // namespace

// =============================================================================================
size_t bias_grad_workspace_bytes(const TView& dy) {
    const size_t npix = (size_t)dy.N * dy.H * dy.W;
    const int TX = pick_tx(dy.C);
    return (size_t)bias_blocks(npix, 256 / TX) * dy.C * sizeof(float);
}

void bias_act_backward(hipStream_t s, const TView& dy, const TView& y, const TView& dz, float* db,
                       int accumulate_db, float* workspace, size_t workspace_bytes) {
    const size_t npix = (size_t)dy.N * dy.H * dy.W;
    const size_t total = npix * dy.C;
    if (!db && y.p && dz.p && plain_contig(dy) && plain_contig(y) && plain_contig(dz) && (total & 3) == 0 &&
        ((((uintptr_t)dy.p) | ((uintptr_t)y.p) | ((uintptr_t)dz.p)) & 15) == 0) {
        ProfScope ps(s, "relu_mask_flat", 0.0, 12.0 * (double)total);
        DL4DS_LAUNCH(relu_mask_flat4_kernel, dim3(ew_blocks(total / 4)), dim3(256), 0, s,
                           reinterpret_cast<const float4*>(dy.p), reinterpret_cast<const float4*>(y.p),
                           reinterpret_cast<float4*>(dz.p), total / 4);
        HIP_CHECK(hipGetLastError());
        return;
    }
    const int TX = pick_tx(dy.C);
    const int TY = 256 / TX;
    const int nb = bias_blocks(npix, TY);
    float* partial = nullptr;
    if (db) {
        DL4DS_REQUIRE(workspace_bytes >= (size_t)nb * dy.C * sizeof(float), "bias grad workspace too small");
        partial = workspace;
    }
    dim3 grid((unsigned)nb, (unsigned)cdiv(dy.C, TX));
    ProfScope ps(s, "bias_act_bwd", 0.0, 4.0 * (double)npix * dy.C * (1 + (y.p ? 1 : 0) + (dz.p ? 1 : 0)));
    switch (TX) {
        case 8: DL4DS_LAUNCH(bias_act_bwd_kernel<8>, grid, dim3(256), 0, s, dy, y, dz, partial, npix); break;
        case 16: DL4DS_LAUNCH(bias_act_bwd_kernel<16>, grid, dim3(256), 0, s, dy, y, dz, partial, npix); break;
        case 32: DL4DS_LAUNCH(bias_act_bwd_kernel<32>, grid, dim3(256), 0, s, dy, y, dz, partial, npix); break;
        default: DL4DS_LAUNCH(bias_act_bwd_kernel<64>, grid, dim3(256), 0, s, dy, y, dz, partial, npix); break;
    }
    HIP_CHECK(hipGetLastError());
    if (db) {
        DL4DS_LAUNCH(reduce_slabs_kernel2, dim3(cdiv(dy.C, 256)), dim3(256), 0, s, partial, db,
                           (size_t)dy.C, nb, accumulate_db);
        HIP_CHECK(hipGetLastError());
    }
}

void view_axpy(hipStream_t s, const TView& src, const TView& dst, float alpha, int accumulate) {
    DL4DS_REQUIRE(src.N == dst.N && src.H == dst.H && src.W == dst.W && src.C == dst.C, "view_axpy: shape mismatch");
    const size_t total = (size_t)src.N * src.H * src.W * src.C;
    if (total == 0) return;
    ProfScope ps(s, "view_axpy", 0.0, 4.0 * (double)total * (2 + (accumulate ? 1 : 0)));
    if (plain_contig(src) && plain_contig(dst) && (total & 3) == 0 && ((((uintptr_t)src.p) | ((uintptr_t)dst.p)) & 15) == 0) {
        DL4DS_LAUNCH(flat_axpy4_kernel, dim3(ew_blocks(total / 4)), dim3(256), 0, s,
                           reinterpret_cast<const float4*>(src.p), reinterpret_cast<float4*>(dst.p), alpha, accumulate, total / 4);
    } else if (src.d2s <= 1 && dst.d2s <= 1 && (src.C & 3) == 0 && (src.ld & 3) == 0 && (dst.ld & 3) == 0 &&
               ((((uintptr_t)src.p) | ((uintptr_t)dst.p)) & 15) == 0 && src.nstride == (size_t)src.H * src.W * src.ld &&
               dst.nstride == (size_t)dst.H * dst.W * dst.ld) {
        const int c4n = src.C / 4;
        const int blocks = ew_blocks(total / 4);
        const size_t stride = (size_t)blocks * 256;
        DL4DS_LAUNCH(strided_axpy4_kernel, dim3(blocks), dim3(256), 0, s, src.p, src.ld, dst.p, dst.ld, c4n,
                           stride / (size_t)c4n, (int)(stride % (size_t)c4n), alpha, accumulate, total / 4);
    } else if (src.d2s <= 1 && dst.d2s <= 1 && !src.sc && !dst.sc && src.nstride == (size_t)src.H * src.W * src.ld &&
               dst.nstride == (size_t)dst.H * dst.W * dst.ld) {
        // channel counts / pitches that are not multiples of four: float2 when everything is even, else one float per thread
        const bool v2 = ((src.C | src.ld | dst.ld) & 1) == 0 && ((((uintptr_t)src.p) | ((uintptr_t)dst.p)) & 7) == 0;
        const int V = v2 ? 2 : 1;
        const int cvn = src.C / V;
        const size_t totalv = total / V;
        const int blocks = ew_blocks(totalv);
        const size_t stride = (size_t)blocks * 256;
        auto kern = v2 ? strided_axpy_small_kernel<2> : strided_axpy_small_kernel<1>;
        DL4DS_LAUNCH(kern, dim3(blocks), dim3(256), 0, s, src.p, src.ld, dst.p, dst.ld, cvn, stride / (size_t)cvn,
                           (int)(stride % (size_t)cvn), alpha, accumulate, totalv);
    } else {
        DL4DS_LAUNCH(view_axpy_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, src, dst, alpha, accumulate, total);
    }
    HIP_CHECK(hipGetLastError());
}

void view_axpy_masked(hipStream_t s, const TView& src, const TView& mask, const TView& dst, int accumulate) {
    if (mask.p == nullptr) { view_axpy(s, src, dst, 1.f, accumulate); return; }
    DL4DS_REQUIRE(src.N == dst.N && src.H == dst.H && src.W == dst.W && src.C == dst.C && mask.C == src.C && mask.N == src.N,
                  "view_axpy_masked: shape mismatch");
    const size_t total = (size_t)src.N * src.H * src.W * src.C;
    if (total == 0) return;
    ProfScope ps(s, "view_axpy_masked", 0.0, 4.0 * (double)total * (3 + (accumulate ? 1 : 0)));
    auto ok4 = [](const TView& v) {
        return v.d2s <= 1 && (v.ld & 3) == 0 && (((uintptr_t)v.p) & 15) == 0 && v.nstride == (size_t)v.H * v.W * v.ld;
    };
    if ((src.C & 3) == 0 && ok4(src) && ok4(mask) && ok4(dst)) {
        const int c4n = src.C / 4;
        const int blocks = ew_blocks(total / 4);
        const size_t stride = (size_t)blocks * 256;
        DL4DS_LAUNCH(strided_masked_axpy4_kernel, dim3(blocks), dim3(256), 0, s, src.p, src.ld, mask.p, mask.ld, dst.p,
                           dst.ld, c4n, stride / (size_t)c4n, (int)(stride % (size_t)c4n), accumulate, total / 4);
    } else {
        DL4DS_LAUNCH(view_masked_axpy_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, src, mask, dst, accumulate, total);
    }
    HIP_CHECK(hipGetLastError());
}

// Concatenate backward in ONE pass: the wide gradient [npx][ld] is read once, contiguously, and every element goes to the dense
// gradient of the input whose channel range holds it (with that input's ReLU mask / accumulation).  Copying slice by slice touches
// every cache line of the wide tensor once per slice (a 2-channel slice of 26 channels reads all of it for 8 % of its bytes).
struct SplitSlice { float* dst; const float* mask; int off, C, acc; };
struct SplitParams { const float* src; int ld, n; SplitSlice sl[4]; };
template <int V>
__global__ void concat_split_kernel(const SplitParams a, int cvn, size_t step_pix, int step_cv, size_t totalv) {
    // (four grid strides per iteration with all loads first measured 7 % slower)
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t pix = e / (size_t)cvn;
    int cv = (int)(e - pix * (size_t)cvn);
    for (; e < totalv; e += (size_t)gridDim.x * blockDim.x) {
        const int c = cv * V;
        float v[V];
        if constexpr (V == 2) { const float2 t = *reinterpret_cast<const float2*>(a.src + pix * (size_t)a.ld + c); v[0] = t.x; v[1] = t.y; }
        else v[0] = a.src[pix * (size_t)a.ld + c];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < a.n && c >= a.sl[k].off && c < a.sl[k].off + a.sl[k].C) {
                const SplitSlice& sl = a.sl[k];
                const size_t o = pix * (size_t)sl.C + (size_t)(c - sl.off);
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    float r = v[j];
                    if (sl.mask) r = sl.mask[o + j] > 0.f ? r : 0.f;
                    if (sl.acc) r += sl.dst[o + j];
                    v[j] = r;
                }
                if constexpr (V == 2) *reinterpret_cast<float2*>(sl.dst + o) = make_float2(v[0], v[1]);
                else sl.dst[o] = v[0];
            }
        }
        pix += step_pix; cv += step_cv;
        if (cv >= cvn) { cv -= cvn; ++pix; }
    }
}

// src: plain [npx][ld]; slices: up to four channel ranges with dense destinations [npx][C_k] (mask likewise, or null).
// Channels of src that belong to no slice are skipped.
void concat_split(hipStream_t s, const float* src, int ld, size_t npx, const ConcatSlice* slices, int n) {
    DL4DS_REQUIRE(n >= 1 && n <= 4, "concat_split: 1..4 slices");
    SplitParams a;
    a.src = src; a.ld = ld; a.n = n;
    bool even = (ld & 1) == 0 && ((uintptr_t)src & 7) == 0;
    double bytes = 4.0 * (double)npx * ld;
    for (int k = 0; k < 4; ++k) {
        if (k < n) {
            a.sl[k] = SplitSlice{slices[k].dst, slices[k].mask, slices[k].off, slices[k].C, slices[k].accumulate};
            even = even && ((slices[k].off | slices[k].C) & 1) == 0 && ((uintptr_t)slices[k].dst & 7) == 0;
            bytes += 4.0 * (double)npx * slices[k].C * (1 + (slices[k].mask ? 1 : 0) + (slices[k].accumulate ? 1 : 0));
        } else {
            a.sl[k] = SplitSlice{nullptr, nullptr, 0, 0, 0};
        }
    }
    const int V = even ? 2 : 1;
    const int cvn = ld / V;
    const size_t totalv = npx * (size_t)cvn;
    if (totalv == 0) return;
    ProfScope ps(s, "concat_split", 0.0, bytes);
    const int blocks = ew_blocks(totalv);
    const size_t stride = (size_t)blocks * 256;
    auto kern = even ? concat_split_kernel<2> : concat_split_kernel<1>;
    DL4DS_LAUNCH(kern, dim3(blocks), dim3(256), 0, s, a, cvn, stride / (size_t)cvn, (int)(stride % (size_t)cvn), totalv);
    HIP_CHECK(hipGetLastError());
}

// Concatenate forward in ONE pass (the mirror of concat_split): the wide tensor [npx][ld] is WRITTEN once, contiguously, every
// element taken from the dense input whose channel range holds it.  Copying input by input writes every 32-byte sector of the
// wide tensor in pieces (a 26-channel pixel is 104 bytes: no slice of it is sector-aligned): 2-channel and 8-channel slices of
// cfg4's concatenation took 273 + 477 us for 335 MB.  Channels that belong to no slice (inputs that live in the buffer
// already) are left alone.
struct JoinSlice { const float* src; int off, C; };
struct JoinParams { float* dst; int ld, n; JoinSlice sl[4]; };
template <int V>
__global__ void concat_join_kernel(const JoinParams a, int cvn, size_t step_pix, int step_cv, size_t totalv) {
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t pix = e / (size_t)cvn;
    int cv = (int)(e - pix * (size_t)cvn);
    for (; e < totalv; e += (size_t)gridDim.x * blockDim.x) {
        const int c = cv * V;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < a.n && c >= a.sl[k].off && c < a.sl[k].off + a.sl[k].C) {
                const float* sp = a.sl[k].src + pix * (size_t)a.sl[k].C + (size_t)(c - a.sl[k].off);
                float* dp = a.dst + pix * (size_t)a.ld + c;
                if constexpr (V == 2) *reinterpret_cast<float2*>(dp) = *reinterpret_cast<const float2*>(sp);
                else dp[0] = sp[0];
            }
        }
        pix += step_pix; cv += step_cv;
        if (cv >= cvn) { cv -= cvn; ++pix; }
    }
}
void concat_join(hipStream_t s, float* dst, int ld, size_t npx, const ConcatSlice* slices, int n) {
    DL4DS_REQUIRE(n >= 1 && n <= 4, "concat_join: 1..4 slices");
    JoinParams a;
    a.dst = dst; a.ld = ld; a.n = n;
    bool even = (ld & 1) == 0 && ((uintptr_t)dst & 7) == 0;
    double bytes = 0;
    for (int k = 0; k < 4; ++k) {
        if (k < n) {
            a.sl[k] = JoinSlice{slices[k].dst, slices[k].off, slices[k].C};
            even = even && ((slices[k].off | slices[k].C) & 1) == 0 && ((uintptr_t)slices[k].dst & 7) == 0;
            bytes += 8.0 * (double)npx * slices[k].C;
        } else {
            a.sl[k] = JoinSlice{nullptr, 0, 0};
        }
    }
    const int V = even ? 2 : 1;
    const int cvn = ld / V;
    const size_t totalv = npx * (size_t)cvn;
    if (totalv == 0) return;
    ProfScope ps(s, "concat_join", 0.0, bytes);
    const int blocks = ew_blocks(totalv);
    const size_t stride = (size_t)blocks * 256;
    auto kern = even ? concat_join_kernel<2> : concat_join_kernel<1>;
    DL4DS_LAUNCH(kern, dim3(blocks), dim3(256), 0, s, a, cvn, stride / (size_t)cvn, (int)(stride % (size_t)cvn), totalv);
    HIP_CHECK(hipGetLastError());
}

bool masked_axpy_pair(hipStream_t s, const float* dy, const float* ya, float* da, int acc_a, const float* yb, float* db, int acc_b, size_t n) {
    if (n == 0 || (n & 3) || ((((uintptr_t)dy) | ((uintptr_t)ya) | ((uintptr_t)da) | ((uintptr_t)yb) | ((uintptr_t)db)) & 15) || da == db) return false;
    ProfScope ps(s, "masked_axpy", 0.0, 4.0 * (double)n * (5 + (acc_a ? 1 : 0) + (acc_b ? 1 : 0)));
    DL4DS_LAUNCH(masked_axpy4_pair_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, s, reinterpret_cast<const float4*>(dy),
                 reinterpret_cast<const float4*>(ya), reinterpret_cast<float4*>(da), reinterpret_cast<const float4*>(yb), reinterpret_cast<float4*>(db),
                 n / 4, acc_a, acc_b);
    HIP_CHECK(hipGetLastError());
    return true;
}

void masked_axpy(hipStream_t s, const float* dy, const float* y, float* dst, size_t n, int accumulate) {
    if (n == 0) return;
    ProfScope ps(s, "masked_axpy", 0.0, 4.0 * (double)n * (3 + (accumulate ? 1 : 0)));
    if ((n & 3) == 0 && ((((uintptr_t)dy) | ((uintptr_t)y) | ((uintptr_t)dst)) & 15) == 0)
        DL4DS_LAUNCH(masked_axpy4_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, s, reinterpret_cast<const float4*>(dy),
                           reinterpret_cast<const float4*>(y), reinterpret_cast<float4*>(dst), n / 4, accumulate);
    else
        DL4DS_LAUNCH(masked_axpy1_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, dy, y, dst, n, accumulate);
    HIP_CHECK(hipGetLastError());
}

void add_act(hipStream_t s, const float* a, const float* b, float* out, size_t n, int relu) {
    ProfScope ps(s, "add_act", 0.0, 12.0 * (double)n);
    DL4DS_LAUNCH(add_act_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, a, b, out, n, relu);
    HIP_CHECK(hipGetLastError());
}
void act_forward(hipStream_t s, const float* x, float* y, size_t n, int kind) {
    ProfScope ps(s, "act_fwd", 0.0, 8.0 * (double)n);
    DL4DS_LAUNCH(act_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, x, y, n, kind);
    HIP_CHECK(hipGetLastError());
}
void act_backward(hipStream_t s, const float* x, const float* dy, float* dx, size_t n, int kind, int accumulate) {
    ProfScope ps(s, "act_bwd", 0.0, 12.0 * (double)n);
    DL4DS_LAUNCH(act_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, x, dy, dx, n, kind, accumulate);
    HIP_CHECK(hipGetLastError());
}
void fill(hipStream_t s, float* p, size_t n, float v) {
    if (n == 0) return;
    ProfScope ps(s, "fill", 0.0, 4.0 * (double)n);
    DL4DS_LAUNCH(fill_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, p, n, v);
    HIP_CHECK(hipGetLastError());
}

void depth_to_space(hipStream_t s, const float* x, float* y, int N, int H, int W, int C, int r) {
    TView src = make_view(const_cast<float*>(x), N, H, W, C);
    TView dst = make_view_d2s(y, N, H, W, C, r);
    view_axpy(s, src, dst, 1.f, 0);
}
void space_to_depth(hipStream_t s, const float* y, float* x, int N, int H, int W, int C, int r) {
    TView src = make_view_d2s(const_cast<float*>(y), N, H, W, C, r);
    TView dst = make_view(x, N, H, W, C);
    view_axpy(s, src, dst, 1.f, 0);
}

// the same per channel QUAD (views whose channel slices are float4-loadable and not depth_to_space stores): one index
// decomposition and 16-byte accesses per four channels instead of per float (cfg5: maxpool2_bwd 0.33 ms per step at 2.8 TB/s)
__global__ void maxpool2_fwd4_kernel(TView x, TView y, size_t total4) {
    const int C4 = y.C >> 2;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C4) * 4;
        int n, oy, ox;
        unflatten_pix(y, e / C4, n, oy, ox);
        const float4 a = *reinterpret_cast<const float4*>(x.p + view_off(x, n, 2 * oy, 2 * ox, c));
        const float4 b = *reinterpret_cast<const float4*>(x.p + view_off(x, n, 2 * oy, 2 * ox + 1, c));
        const float4 d = *reinterpret_cast<const float4*>(x.p + view_off(x, n, 2 * oy + 1, 2 * ox, c));
        const float4 f = *reinterpret_cast<const float4*>(x.p + view_off(x, n, 2 * oy + 1, 2 * ox + 1, c));
        float4 m;
        m.x = fmaxf(fmaxf(fmaxf(a.x, b.x), d.x), f.x); m.y = fmaxf(fmaxf(fmaxf(a.y, b.y), d.y), f.y);
        m.z = fmaxf(fmaxf(fmaxf(a.z, b.z), d.z), f.z); m.w = fmaxf(fmaxf(fmaxf(a.w, b.w), d.w), f.w);
        *reinterpret_cast<float4*>(y.p + view_off(y, n, oy, ox, c)) = m;
    }
}
__global__ void maxpool2_bwd4_kernel(TView x, TView y, TView dy, TView dx, int accumulate, int relu_mask, size_t total4) {
    const int C4 = y.C >> 2;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C4) * 4;
        int n, oy, ox;
        unflatten_pix(y, e / C4, n, oy, ox);
        const float4 m4 = *reinterpret_cast<const float4*>(y.p + view_off(y, n, oy, ox, c));
        const float4 g4 = *reinterpret_cast<const float4*>(dy.p + view_off(dy, n, oy, ox, c));
        const float m[4] = {m4.x, m4.y, m4.z, m4.w}, g[4] = {g4.x, g4.y, g4.z, g4.w};
        float4 xv4[4], old4[4];
        size_t o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int iy = 2 * oy + (k >> 1), ix = 2 * ox + (k & 1);
            xv4[k] = *reinterpret_cast<const float4*>(x.p + view_off(x, n, iy, ix, c));
            o[k] = view_off(dx, n, iy, ix, c);
            if (accumulate) old4[k] = *reinterpret_cast<const float4*>(dx.p + o[k]);
        }
        bool found[4] = {false, false, false, false};
#pragma unroll
        for (int k = 0; k < 4; ++k) {                    // (window order as in the scalar kernel: the first maximum takes the gradient)
            const float xv[4] = {xv4[k].x, xv4[k].y, xv4[k].z, xv4[k].w};
            float gv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                gv[j] = 0.f;
                if (!found[j] && xv[j] == m[j]) { gv[j] = g[j]; found[j] = true; }
                if (relu_mask && !(xv[j] > 0.f)) gv[j] = 0.f;
            }
            float4 r = make_float4(gv[0], gv[1], gv[2], gv[3]);
            if (accumulate) { r.x += old4[k].x; r.y += old4[k].y; r.z += old4[k].z; r.w += old4[k].w; }
            *reinterpret_cast<float4*>(dx.p + o[k]) = r;
        }
    }
}
static bool pool_quad_ok(const TView& v) { return v.vec && v.d2s <= 1 && (v.C & 3) == 0 && !v.sc; }

void maxpool2_forward(hipStream_t s, const TView& x, const TView& y) {
    DL4DS_REQUIRE(y.H == x.H / 2 && y.W == x.W / 2 && y.C == x.C && y.N == x.N, "maxpool2: shapes");
    const size_t total = (size_t)y.N * y.H * y.W * y.C;
    ProfScope ps(s, "maxpool2_fwd", 0.0, 4.0 * (double)total * 5);
    static const bool no_quad = exp_env("DL4DS_NO_POOL_QUAD") != nullptr;      // (A/B)
    if (!no_quad && pool_quad_ok(x) && pool_quad_ok(y)) {
        DL4DS_LAUNCH(maxpool2_fwd4_kernel, dim3(ew_blocks(total / 4)), dim3(256), 0, s, x, y, total / 4);
        HIP_CHECK(hipGetLastError());
        return;
    }
    DL4DS_LAUNCH(maxpool2_fwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, x, y, total);
    HIP_CHECK(hipGetLastError());
}
void maxpool2_backward(hipStream_t s, const TView& x, const TView& y, const TView& dy, const TView& dx, int accumulate,
                       int relu_mask) {
    DL4DS_REQUIRE((x.H % 2 == 0 && x.W % 2 == 0) || accumulate,
                  "maxpool2 backward with odd sizes needs a pre-zeroed accumulate target");
    const size_t total = (size_t)y.N * y.H * y.W * y.C;
    ProfScope ps(s, "maxpool2_bwd", 0.0, 4.0 * (double)total * (2 + 4 + 4 + (accumulate ? 4 : 0)));
    static const bool no_quad = exp_env("DL4DS_NO_POOL_QUAD") != nullptr;      // (A/B)
    if (!no_quad && pool_quad_ok(x) && pool_quad_ok(y) && pool_quad_ok(dy) && pool_quad_ok(dx)) {
        DL4DS_LAUNCH(maxpool2_bwd4_kernel, dim3(ew_blocks(total / 4)), dim3(256), 0, s, x, y, dy, dx, accumulate, relu_mask, total / 4);
        HIP_CHECK(hipGetLastError());
        return;
    }
    DL4DS_LAUNCH(maxpool2_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, x, y, dy, dx, accumulate, relu_mask, total);
    HIP_CHECK(hipGetLastError());
}

void resize_bilinear_forward(hipStream_t s, const TView& x, const TView& y) {
    const size_t total = (size_t)y.N * y.H * y.W * y.C;
    ProfScope ps(s, "resize_bilinear_fwd", 0.0, 4.0 * ((double)total + (double)x.N * x.H * x.W * x.C));
    DL4DS_LAUNCH(resize_fwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, x, y, (float)x.H / (float)y.H,
                       (float)x.W / (float)y.W, total);
    HIP_CHECK(hipGetLastError());
}
void resize_bilinear_backward(hipStream_t s, const TView& dy, const TView& dx, int accumulate) {
    const size_t total = (size_t)dx.N * dx.H * dx.W * dx.C;
    ProfScope ps(s, "resize_bilinear_bwd", 0.0, 4.0 * ((double)total * (accumulate ? 2 : 1) + (double)dy.N * dy.H * dy.W * dy.C));
    DL4DS_LAUNCH(resize_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, dy, dx, (float)dx.H / (float)dy.H,
                       (float)dx.W / (float)dy.W, accumulate, total);
    HIP_CHECK(hipGetLastError());
}

void resize_nearest_forward(hipStream_t s, const TView& x, const TView& y) {
    const size_t total = (size_t)y.N * y.H * y.W * y.C;
    ProfScope ps(s, "resize_nearest_fwd", 0.0, 4.0 * ((double)total + (double)x.N * x.H * x.W * x.C));
    DL4DS_LAUNCH(resize_nearest_fwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, x, y, (float)x.H / (float)y.H,
                       (float)x.W / (float)y.W, total);
    HIP_CHECK(hipGetLastError());
}
void resize_nearest_backward(hipStream_t s, const TView& dy, const TView& dx, int accumulate) {
    const size_t total = (size_t)dx.N * dx.H * dx.W * dx.C;
    ProfScope ps(s, "resize_nearest_bwd", 0.0, 4.0 * ((double)total * (accumulate ? 2 : 1) + (double)dy.N * dy.H * dy.W * dy.C));
    DL4DS_LAUNCH(resize_nearest_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, dy, dx, (float)dx.H / (float)dy.H,
                       (float)dx.W / (float)dy.W, accumulate, total);
    HIP_CHECK(hipGetLastError());
}

void localconv_forward(hipStream_t s, const TView& x, const float* w, const float* b, const TView& y) {
    DL4DS_REQUIRE(x.C <= kLcMax && y.C <= kLcMax, "localconv: at most 8 channels in/out");
    const size_t npix = (size_t)x.N * x.H * x.W;
    ProfScope ps(s, "localconv_fwd", 2.0 * npix * x.C * y.C, 4.0 * ((double)npix * (x.C + y.C) + (double)x.H * x.W * (x.C + 1) * y.C));
    DL4DS_LAUNCH(localconv_fwd_kernel, dim3(ew_blocks(npix)), dim3(256), 0, s, x, w, b, y, npix);
    HIP_CHECK(hipGetLastError());
}
void localconv_backward(hipStream_t s, const TView& x, const float* w, const TView& dy, const TView& dx,
                        int accumulate_dx, float* dw, float* db, int accumulate_dw) {
    DL4DS_REQUIRE(x.C <= kLcMax && dy.C <= kLcMax, "localconv: at most 8 channels in/out");
    const size_t nhw = (size_t)x.H * x.W;
    ProfScope ps(s, "localconv_bwd", 4.0 * nhw * x.N * x.C * dy.C,
                 4.0 * ((double)nhw * x.N * (2 * x.C + dy.C) + 2.0 * (double)nhw * (x.C + 1) * dy.C));
    const dim3 grid((unsigned)cdivz(nhw, 64));
    if (x.C == 2 && dy.C == 2)      // LocalizedConvBlock (blocks.py:312-336): TransitionBlock(2) -> LocallyConnected2D(2)
        DL4DS_LAUNCH((localconv_bwd_kernel<2, 2>), grid, dim3(256), 0, s, x, w, dy, dx, accumulate_dx, dw, db, accumulate_dw);
    else
        DL4DS_LAUNCH((localconv_bwd_kernel<0, 0>), grid, dim3(256), 0, s, x, w, dy, dx, accumulate_dx, dw, db, accumulate_dw);
    HIP_CHECK(hipGetLastError());
}


// ---------------------------------------------------------------------------------------------
// tf.repeat(tf.expand_dims(s, 1), T, axis=1) and its gradient (spt_postups.py:139-140) as ONE launch each: the op used to
// issue B*T copies forward and B*T dependent accumulations backward (128 + 128 launches of a few microseconds in cfg4).
__global__ void repeat_time_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, size_t ps, int T, size_t total) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t b = e / ps, r = e - b * ps;
        const float v = in[e];
        float* o = out + b * T * ps + r;
        for (int t = 0; t < T; ++t) o[(size_t)t * ps] = v;
    }
}
__global__ void repeat_time_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din, size_t ps, int T, size_t total,
                                       int accumulate) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t b = e / ps, r = e - b * ps;
        const float* o = dout + b * T * ps + r;
        float s = accumulate ? din[e] : 0.f;
        for (int t = 0; t < T; ++t) s += o[(size_t)t * ps];      // (same order as the T accumulating copies it replaces)
        din[e] = s;
    }
}
void repeat_time_forward(hipStream_t s, const float* in, float* out, int B, int T, size_t ps) {
    const size_t total = (size_t)B * ps;
    if (total == 0) return;
    ProfScope pp(s, "repeat_time_fwd", 0.0, 4.0 * (double)total * (1 + T));
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
    DL4DS_LAUNCH(repeat_time_fwd_kernel, dim3(blocks), dim3(256), 0, s, in, out, ps, T, total);
    HIP_CHECK(hipGetLastError());
}
// ... straight into a channel slice of a wider buffer (the Concatenate that follows, GTensor::alias_of): out is a view with pixel
// pitch ld >= C; V floats per access (4 / 2 / 1: what the slice's offset and pitch allow)
template <int V>
__global__ void repeat_time_fwd_view_kernel(const float* __restrict__ in, float* __restrict__ out, int cvn, size_t hw, int C, int ld,
                                            size_t nstride, int T, size_t totalv) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < totalv; e += (size_t)gridDim.x * blockDim.x) {
        const size_t px = e / (size_t)cvn;                    // pixel index over (b, y, x)
        const int cv = (int)(e - px * (size_t)cvn);
        const size_t b = px / hw, r = px - b * hw;
        const float* ip = in + px * (size_t)C + (size_t)cv * V;
        float* o = out + (b * T) * nstride + r * (size_t)ld + (size_t)cv * V;
        if constexpr (V == 4) {
            const float4 v = *reinterpret_cast<const float4*>(ip);
            for (int t = 0; t < T; ++t) *reinterpret_cast<float4*>(o + (size_t)t * nstride) = v;
        } else if constexpr (V == 2) {
            const float2 v = *reinterpret_cast<const float2*>(ip);
            for (int t = 0; t < T; ++t) *reinterpret_cast<float2*>(o + (size_t)t * nstride) = v;
        } else {
            const float v = ip[0];
            for (int t = 0; t < T; ++t) o[(size_t)t * nstride] = v;
        }
    }
}
void repeat_time_forward_view(hipStream_t s, const float* in, const TView& out, int B, int T) {
    DL4DS_REQUIRE(out.d2s <= 1 && out.N == B * T, "repeat_time: plain output view of B * T frames expected");
    const size_t hw = (size_t)out.H * out.W, total = (size_t)B * hw * out.C;
    if (total == 0) return;
    ProfScope pp(s, "repeat_time_fwd", 0.0, 4.0 * (double)total * (1 + T));
    const uintptr_t a = (uintptr_t)out.p | (uintptr_t)in;
    const int V = ((out.C & 3) == 0 && (out.ld & 3) == 0 && (a & 15) == 0) ? 4 : (((out.C & 1) == 0 && (out.ld & 1) == 0 && (a & 7) == 0) ? 2 : 1);
    const size_t totalv = total / V;
    const int blocks = (int)std::min<size_t>((totalv + 255) / 256, 8192);
    const int cvn = out.C / V;
    if (V == 4) DL4DS_LAUNCH(repeat_time_fwd_view_kernel<4>, dim3(blocks), dim3(256), 0, s, in, out.p, cvn, hw, out.C, out.ld, out.nstride, T, totalv);
    else if (V == 2) DL4DS_LAUNCH(repeat_time_fwd_view_kernel<2>, dim3(blocks), dim3(256), 0, s, in, out.p, cvn, hw, out.C, out.ld, out.nstride, T, totalv);
    else DL4DS_LAUNCH(repeat_time_fwd_view_kernel<1>, dim3(blocks), dim3(256), 0, s, in, out.p, cvn, hw, out.C, out.ld, out.nstride, T, totalv);
    HIP_CHECK(hipGetLastError());
}
// ... and its gradient read from a channel slice of the Concatenate's gradient (GTensor::galias): same summation order
__global__ void repeat_time_bwd_view_kernel(const float* __restrict__ dout, float* __restrict__ din, size_t hw, int C, int ld,
                                            size_t nstride, int T, size_t total, int accumulate) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t px = e / (size_t)C;
        const int c = (int)(e - px * (size_t)C);
        const size_t b = px / hw, r = px - b * hw;
        const float* o = dout + (b * T) * nstride + r * (size_t)ld + c;
        float sm = accumulate ? din[e] : 0.f;
        for (int t = 0; t < T; ++t) sm += o[(size_t)t * nstride];
        din[e] = sm;
    }
}
void repeat_time_backward_view(hipStream_t s, const TView& dout, float* din, int B, int T, int accumulate) {
    DL4DS_REQUIRE(dout.d2s <= 1 && dout.N == B * T, "repeat_time: plain gradient view of B * T frames expected");
    const size_t hw = (size_t)dout.H * dout.W, total = (size_t)B * hw * dout.C;
    if (total == 0) return;
    ProfScope pp(s, "repeat_time_bwd", 0.0, 4.0 * (double)total * (1 + T + (accumulate ? 1 : 0)));
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 8192);
    DL4DS_LAUNCH(repeat_time_bwd_view_kernel, dim3(blocks), dim3(256), 0, s, dout.p, din, hw, dout.C, dout.ld, dout.nstride, T, total,
                       accumulate);
    HIP_CHECK(hipGetLastError());
}
void repeat_time_backward(hipStream_t s, const float* dout, float* din, int B, int T, size_t ps, int accumulate) {
    const size_t total = (size_t)B * ps;
    if (total == 0) return;
    ProfScope pp(s, "repeat_time_bwd", 0.0, 4.0 * (double)total * (1 + T + (accumulate ? 1 : 0)));
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
    DL4DS_LAUNCH(repeat_time_bwd_kernel, dim3(blocks), dim3(256), 0, s, dout, din, ps, T, total, accumulate);
    HIP_CHECK(hipGetLastError());
}


// ---------------------------------------------------------------------------------------------
// Resizing(..., interpolation='bicubic') = tf.image.resize(method='bicubic') = the ResizeBicubic op with half-pixel centres
// (blocks.py:473-489): Keys cubic, A = -0.5, weights taken from a 1024-step table at lrintf(frac * 1024), taps outside the
// image dropped and the rest renormalised.  Both passes are table driven and deterministic: forward gathers 4 x 4 taps per
// output element; backward gathers, per INPUT element, the (output index, weight) pairs that reference it (CSR built on the
// host once per op) -- the exact transpose, no atomics.
__global__ void resize_table_fwd_kernel(TView x, TView y, const int* __restrict__ iy, const float* __restrict__ wy,
                                        const int* __restrict__ ix, const float* __restrict__ wx, int ky, int kx, size_t total) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % y.C);
        size_t r = e / y.C;
        const int xo = (int)(r % y.W); r /= y.W;
        const int yo = (int)(r % y.H);
        const int n = (int)(r / y.H);
        float acc = 0.f;
        for (int a = 0; a < ky; ++a) {
            float row = 0.f;
            for (int b = 0; b < kx; ++b) row += wx[xo * kx + b] * x.p[view_off(x, n, iy[yo * ky + a], ix[xo * kx + b], c)];
            acc += wy[yo * ky + a] * row;
        }
        y.p[view_off(y, n, yo, xo, c)] = acc;
    }
}
__global__ void resize_table_bwd_kernel(TView dy, TView dx, const int* __restrict__ py, const int* __restrict__ oy,
                                        const float* __restrict__ vy, const int* __restrict__ px, const int* __restrict__ ox,
                                        const float* __restrict__ vx, int accumulate, size_t total) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % dx.C);
        size_t r = e / dx.C;
        const int xi = (int)(r % dx.W); r /= dx.W;
        const int yi = (int)(r % dx.H);
        const int n = (int)(r / dx.H);
        float acc = 0.f;
        for (int a = py[yi]; a < py[yi + 1]; ++a) {
            float row = 0.f;
            for (int b = px[xi]; b < px[xi + 1]; ++b) row += vx[b] * dy.p[view_off(dy, n, oy[a], ox[b], c)];
            acc += vy[a] * row;
        }
        const size_t o = view_off(dx, n, yi, xi, c);
        dx.p[o] = accumulate ? dx.p[o] + acc : acc;
    }
}
// float4 over the channels (both views float4-loadable): the index / weight lookups are shared by four channels
__global__ void resize_table_fwd4_kernel(TView x, TView y, const int* __restrict__ iy, const float* __restrict__ wy,
                                         const int* __restrict__ ix, const float* __restrict__ wx, int ky, int kx, size_t total4) {
    const int C4 = y.C >> 2;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += (size_t)gridDim.x * blockDim.x) {
        // (32-bit index arithmetic: the launcher checks total4 < 2^32; 64-bit divisions cost ~40 instructions each)
        const unsigned e32 = (unsigned)e;
        unsigned r = e32 / (unsigned)C4;
        const int c = (int)(e32 - r * (unsigned)C4) * 4;
        const unsigned r1 = r / (unsigned)y.W;
        const int xo = (int)(r - r1 * (unsigned)y.W);
        const int n = (int)(r1 / (unsigned)y.H);
        const int yo = (int)(r1 - (unsigned)n * (unsigned)y.H);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int a = 0; a < ky; ++a) {
            float4 row = make_float4(0.f, 0.f, 0.f, 0.f);
            const int sy = iy[yo * ky + a];
            for (int b = 0; b < kx; ++b) {
                const float w = wx[xo * kx + b];
                const float4 v = *reinterpret_cast<const float4*>(x.p + view_off(x, n, sy, ix[xo * kx + b], c));
                row.x += w * v.x; row.y += w * v.y; row.z += w * v.z; row.w += w * v.w;
            }
            const float w = wy[yo * ky + a];
            acc.x += w * row.x; acc.y += w * row.y; acc.z += w * row.z; acc.w += w * row.w;
        }
        *reinterpret_cast<float4*>(y.p + view_off(y, n, yo, xo, c)) = acc;
    }
}
// K x K taps known at compile time (bilinear: 2, bicubic: 4), plain views: the 2 K index / weight loads and then the K * K data
// loads are all issued before the arithmetic (the generic loop's loads depend on each other tap by tap: 278 us for the 537 MB
// output of cfg4's 64^2 -> 256^2 resize)
template <int K>
__global__ void resize_table_fwdk_kernel(const float* __restrict__ xp, float* __restrict__ yp, int Hi, int Wi, int Ho, int Wo, int C,
                                         const int* __restrict__ iy, const float* __restrict__ wy, const int* __restrict__ ix,
                                         const float* __restrict__ wx, int rows, unsigned m_c4) {
    // a block takes whole output rows: image, output row and the row taps are wave-uniform (scalar), a thread only splits its
    // position in the row into (column, channel quad) -- the flat form spent three run-time integer divisions per float4 and was
    // bound by them (0.20 ms for cfg4's 537 MB output, 2.8 TB/s)
    const int C4 = C >> 2, per_row = Wo * C4;
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const int n = row / Ho, yo = row - n * Ho;
        int sy[K];
        float fy[K];
#pragma unroll
        for (int a = 0; a < K; ++a) { sy[a] = iy[yo * K + a]; fy[a] = wy[yo * K + a]; }
        const float* img = xp + (size_t)n * Hi * Wi * C;
        float* out = yp + (size_t)row * per_row * 4;
        for (int e = threadIdx.x; e < per_row; e += blockDim.x) {
            const int xo = fast_div(e, m_c4);
            const int c = (e - xo * C4) * 4;
            int sx[K];
            float fx[K];
#pragma unroll
            for (int b = 0; b < K; ++b) { sx[b] = ix[xo * K + b]; fx[b] = wx[xo * K + b]; }
            float4 v[K][K];
#pragma unroll
            for (int a = 0; a < K; ++a)
#pragma unroll
                for (int b = 0; b < K; ++b) v[a][b] = *reinterpret_cast<const float4*>(img + ((size_t)sy[a] * Wi + sx[b]) * C + c);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int a = 0; a < K; ++a) {                  // (the generic kernel's order of operations)
                float4 rowv = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int b = 0; b < K; ++b) { rowv.x += fx[b] * v[a][b].x; rowv.y += fx[b] * v[a][b].y; rowv.z += fx[b] * v[a][b].z; rowv.w += fx[b] * v[a][b].w; }
                acc.x += fy[a] * rowv.x; acc.y += fy[a] * rowv.y; acc.z += fy[a] * rowv.z; acc.w += fy[a] * rowv.w;
            }
            *reinterpret_cast<float4*>(out + (size_t)e * 4) = acc;
        }
    }
}
__global__ void resize_table_bwd4_kernel(TView dy, TView dx, const int* __restrict__ py, const int* __restrict__ oy,
                                         const float* __restrict__ vy, const int* __restrict__ px, const int* __restrict__ ox,
                                         const float* __restrict__ vx, int accumulate, size_t total4) {
    const int C4 = dx.C >> 2;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += (size_t)gridDim.x * blockDim.x) {
        const unsigned e32 = (unsigned)e;
        unsigned r = e32 / (unsigned)C4;
        const int c = (int)(e32 - r * (unsigned)C4) * 4;
        const unsigned r1 = r / (unsigned)dx.W;
        const int xi = (int)(r - r1 * (unsigned)dx.W);
        const int n = (int)(r1 / (unsigned)dx.H);
        const int yi = (int)(r1 - (unsigned)n * (unsigned)dx.H);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int a = py[yi]; a < py[yi + 1]; ++a) {
            float4 row = make_float4(0.f, 0.f, 0.f, 0.f);
            const int sy = oy[a];
            for (int b = px[xi]; b < px[xi + 1]; ++b) {
                const float w = vx[b];
                const float4 v = *reinterpret_cast<const float4*>(dy.p + view_off(dy, n, sy, ox[b], c));
                row.x += w * v.x; row.y += w * v.y; row.z += w * v.z; row.w += w * v.w;
            }
            const float w = vy[a];
            acc.x += w * row.x; acc.y += w * row.y; acc.z += w * row.z; acc.w += w * row.w;
        }
        float4* d = reinterpret_cast<float4*>(dx.p + view_off(dx, n, yi, xi, c));
        if (accumulate) { const float4 o = *d; acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
        *d = acc;
    }
}
// ... with the row of an input pixel's contributions as MT INDEPENDENT loads (MT >= the longest row of the transposed x table;
// entries past a row's end repeat its last column with weight 0: same summation order, the padding adds zeros).  The loop above
// has run-time bounds: one load in flight per thread, 64 of them one after the other for a x4 bilinear resize (cfg4: 0.31 ms
// at 1.9 TB/s).
template <int MT>
__global__ void resize_table_bwd4u_kernel(TView dy, TView dx, const int* __restrict__ py, const int* __restrict__ oy,
                                          const float* __restrict__ vy, const int* __restrict__ px, const int* __restrict__ ox,
                                          const float* __restrict__ vx, int accumulate, size_t total4) {
    const int C4 = dx.C >> 2;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += (size_t)gridDim.x * blockDim.x) {
        const unsigned e32 = (unsigned)e;
        unsigned r = e32 / (unsigned)C4;
        const int c = (int)(e32 - r * (unsigned)C4) * 4;
        const unsigned r1 = r / (unsigned)dx.W;
        const int xi = (int)(r - r1 * (unsigned)dx.W);
        const int n = (int)(r1 / (unsigned)dx.H);
        const int yi = (int)(r1 - (unsigned)n * (unsigned)dx.H);
        const int b0 = px[xi], nb = px[xi + 1] - b0;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (nb > 0) {
            int sx[MT];
            float wx[MT];
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const int b = b0 + min(j, nb - 1);
                sx[j] = ox[b];
                wx[j] = j < nb ? vx[b] : 0.f;
            }
            for (int a = py[yi]; a < py[yi + 1]; ++a) {
                const int sy = oy[a];
                float4 v[MT];
#pragma unroll
                for (int j = 0; j < MT; ++j) v[j] = *reinterpret_cast<const float4*>(dy.p + view_off(dy, n, sy, sx[j], c));
                float4 row = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int j = 0; j < MT; ++j) { row.x += wx[j] * v[j].x; row.y += wx[j] * v[j].y; row.z += wx[j] * v[j].z; row.w += wx[j] * v[j].w; }
                const float w = vy[a];
                acc.x += w * row.x; acc.y += w * row.y; acc.z += w * row.z; acc.w += w * row.w;
            }
        }
        float4* d = reinterpret_cast<float4*>(dx.p + view_off(dx, n, yi, xi, c));
        if (accumulate) { const float4 o = *d; acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
        *d = acc;
    }
}
void resize_table_forward(hipStream_t s, const TView& x, const TView& y, const int* iy, const float* wy, const int* ix, const float* wx,
                          int ky, int kx) {
    const size_t total = (size_t)y.N * y.H * y.W * y.C;
    auto dense = [](const TView& v) { return v.d2s <= 1 && v.ld == v.C && v.nstride == (size_t)v.H * v.W * v.C && v.vec; };
    const size_t rows = (size_t)y.N * y.H, per_row4 = (size_t)y.W * (y.C >> 2);
    if (dense(x) && dense(y) && !x.sc && ky == kx && (ky == 2 || ky == 4) && rows < (1ull << 31) && per_row4 < (1u << 20) && (y.C >> 2) <= 4096 &&
        !exp_env("DL4DS_NO_RESIZE_FWDK")) {
        ProfScope ps(s, "resize_table_fwd", 0.0, 4.0 * ((double)total + (double)x.N * x.H * x.W * x.C));
        auto kern = ky == 2 ? resize_table_fwdk_kernel<2> : resize_table_fwdk_kernel<4>;
        const int threads = per_row4 >= 256 ? 256 : 64;
        DL4DS_LAUNCH(kern, dim3((unsigned)std::min<size_t>(rows, 65536)), dim3(threads), 0, s, x.p, y.p, x.H, x.W, y.H, y.W, y.C, iy, wy, ix, wx,
                     (int)rows, div_magic(y.C >> 2));
        HIP_CHECK(hipGetLastError());
        return;
    }
    if (x.vec && y.vec && x.d2s <= 1 && y.d2s <= 1 && !x.sc && total / 4 < (1ull << 32)) {
        ProfScope ps(s, "resize_table_fwd", 0.0, 4.0 * ((double)total + (double)x.N * x.H * x.W * x.C));
        DL4DS_LAUNCH(resize_table_fwd4_kernel, dim3(ew_blocks(total / 4)), dim3(256), 0, s, x, y, iy, wy, ix, wx, ky, kx, total / 4);
        HIP_CHECK(hipGetLastError());
        return;
    }
    ProfScope ps(s, "resize_table_fwd", 0.0, 4.0 * ((double)total + (double)x.N * x.H * x.W * x.C));
    DL4DS_LAUNCH(resize_table_fwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, x, y, iy, wy, ix, wx, ky, kx, total);
    HIP_CHECK(hipGetLastError());
}
void resize_table_backward(hipStream_t s, const TView& dy, const TView& dx, const int* py, const int* oy, const float* vy,
                           const int* px, const int* ox, const float* vx, int accumulate, int max_taps_x) {
    const size_t total = (size_t)dx.N * dx.H * dx.W * dx.C;
    if (dy.vec && dx.vec && dy.d2s <= 1 && dx.d2s <= 1 && !dy.sc && total / 4 < (1ull << 32)) {
        ProfScope ps(s, "resize_table_bwd", 0.0, 4.0 * ((double)total * (accumulate ? 2 : 1) + (double)dy.N * dy.H * dy.W * dy.C));
        static const bool no_u = exp_env("DL4DS_NO_RESIZE_BWDU") != nullptr;      // (A/B)
        if (max_taps_x > 0 && max_taps_x <= 8 && !no_u) {
            auto kern = max_taps_x <= 4 ? resize_table_bwd4u_kernel<4> : resize_table_bwd4u_kernel<8>;
            DL4DS_LAUNCH(kern, dim3(ew_blocks(total / 4)), dim3(256), 0, s, dy, dx, py, oy, vy, px, ox, vx, accumulate, total / 4);
            HIP_CHECK(hipGetLastError());
            return;
        }
        DL4DS_LAUNCH(resize_table_bwd4_kernel, dim3(ew_blocks(total / 4)), dim3(256), 0, s, dy, dx, py, oy, vy, px, ox, vx, accumulate,
                           total / 4);
        HIP_CHECK(hipGetLastError());
        return;
    }
    ProfScope ps(s, "resize_table_bwd", 0.0, 4.0 * ((double)total * (accumulate ? 2 : 1) + (double)dy.N * dy.H * dy.W * dy.C));
    DL4DS_LAUNCH(resize_table_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, dy, dx, py, oy, vy, px, ox, vx, accumulate, total);
    HIP_CHECK(hipGetLastError());
}
