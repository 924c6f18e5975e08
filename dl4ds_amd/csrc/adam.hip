// tf.keras Adam on a flat fp32 parameter arena (supervised.py:353; cgan.py:277-278):
//   m <- b1 m + (1-b1) g ; v <- b2 v + (1-b2) g^2 ; w <- w - lr_t * m / (sqrt(v) + eps)
// with lr_t = lr*sqrt(1-b2^t)/(1-b1^t) folded by the caller and eps OUTSIDE the bias correction.
// grad_scale folds the 1/world_size of the data-parallel gradient average.  HBM-bound: 16 B read +
// 12 B written per parameter, float4 vectorised.
#include "ops.h"
#include "prof.h"
#include <algorithm>

namespace {
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ w, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, size_t n, float lr_t,
                                                   float b1, float b2, float eps, float gs, const unsigned* err) {
    // a step whose kernels raised the sticky device error word (runtime.h) has INVALID gradients; the host only learns of it
    // at its next wait, after this kernel.  Leave parameters and moments untouched in that case (one uncached read per block).
    if (err && __builtin_nontemporal_load(err) != 0u) return;
    const size_t n4 = n >> 2;
    float4* w4 = reinterpret_cast<float4*>(w);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += stride) {
        float4 ww = w4[e], gg = g4[e], mm = m4[e], vv = v4[e];
#define ADAM1(c)                                               \
    {                                                          \
        const float gr = gg.c * gs;                            \
        mm.c = b1 * mm.c + (1.f - b1) * gr;                    \
        vv.c = b2 * vv.c + (1.f - b2) * gr * gr;               \
        ww.c -= lr_t * mm.c / (sqrtf(vv.c) + eps);             \
    }
        ADAM1(x) ADAM1(y) ADAM1(z) ADAM1(w)
#undef ADAM1
        w4[e] = ww; m4[e] = mm; v4[e] = vv;
    }
    for (size_t e = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const float gr = g[e] * gs;
        const float mm = b1 * m[e] + (1.f - b1) * gr;
        const float vv = b2 * v[e] + (1.f - b2) * gr * gr;
        m[e] = mm; v[e] = vv;
        w[e] -= lr_t * mm / (sqrtf(vv) + eps);
    }
}
}  // namespace

void adam_update(hipStream_t s, float* w, const float* g, float* m, float* v, size_t n, float lr_t, float beta1,
                 float beta2, float eps, float grad_scale, const unsigned* err_word) {
    if (n == 0) return;
    ProfScope ps(s, "adam", 0.0, 28.0 * (double)n);
    const int blocks = (int)std::max<size_t>(1, std::min<size_t>(cdivz(n / 4 + 1, 256), 2048));
    DL4DS_LAUNCH(adam_kernel, dim3(blocks), dim3(256), 0, s, w, g, m, v, n, lr_t, beta1, beta2, eps, grad_scale, err_word);
    HIP_CHECK(hipGetLastError());
}
