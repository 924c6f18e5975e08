// Gate arithmetic of tf.keras.layers.ConvLSTM2D (tf.keras-2 defaults: activation=tanh,
// recurrent_activation=hard_sigmoid = clip(0.2x+0.5,0,1), gate order i,f,c,o, zero initial state) --
// dl4ds/models/blocks.py:350-355.  The two convolutions per step (input kernel batched over all T frames,
// recurrent kernel per step) run on the MFMA conv kernels; this file fuses everything between them:
//   forward : z -> (i,f,g,o) -> c_t = f*c_{t-1} + i*g ; h_t = o*tanh(c_t) ; out_t = [relu](h_t)
//   backward: one BPTT step producing dz_t (for the wgrad/dgrad convs) and dc_{t-1}
// HBM-bound: ~10 floats read/written per (pixel, filter).
#include "ops.h"
#include "prof.h"
#include "head.h"
#include <algorithm>

namespace {
__device__ __forceinline__ float hsig(float z) { return fminf(fmaxf(0.2f * z + 0.5f, 0.f), 1.f); }
__device__ __forceinline__ float dhsig(float z) { return (z >= -2.5f && z <= 2.5f) ? 0.2f : 0.f; }

__global__ void gates_fwd_kernel(TView z, TView cp, TView c, TView h, TView out, int relu, int first, size_t total) {
    const int F = c.C;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int f = (int)(e % F);
        size_t pix = e / F;
        const int x = (int)(pix % c.W);
        pix /= c.W;
        const int y = (int)(pix % c.H);
        const int n = (int)(pix / c.H);
        const size_t zo = view_off(z, n, y, x, 0);
        const float i = hsig(z.p[zo + f]);
        const float fg = hsig(z.p[zo + F + f]);
        const float g = tanhf(z.p[zo + 2 * F + f]);
        const float o = hsig(z.p[zo + 3 * F + f]);
        float cc = i * g;
        if (!first) cc += fg * cp.p[view_off(cp, n, y, x, f)];
        const float hh = o * tanhf(cc);
        c.p[view_off(c, n, y, x, f)] = cc;
        if (h.p) h.p[view_off(h, n, y, x, f)] = hh;        // (null at the last step: nobody reads h_{T-1} as a recurrent input)
        out.p[view_off(out, n, y, x, f)] = relu ? fmaxf(hh, 0.f) : hh;
    }
}

__global__ void gates_bwd_kernel(TView z, TView cp, TView c, TView out, TView dout, TView dh_rec, TView dc_next, TView dz,
                                 int relu, int first, int last, size_t total) {
    const int F = c.C;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int f = (int)(e % F);
        size_t pix = e / F;
        const int x = (int)(pix % c.W);
        pix /= c.W;
        const int y = (int)(pix % c.H);
        const int n = (int)(pix / c.H);
        const size_t zo = view_off(z, n, y, x, 0);
        const float zi = z.p[zo + f], zf = z.p[zo + F + f], zc = z.p[zo + 2 * F + f], zq = z.p[zo + 3 * F + f];
        const float i = hsig(zi), fg = hsig(zf), g = tanhf(zc), o = hsig(zq);
        float dh = dout.p[view_off(dout, n, y, x, f)];
        if (relu) dh = (out.p[view_off(out, n, y, x, f)] > 0.f) ? dh : 0.f;
        const size_t ro = view_off(dh_rec, n, y, x, f);
        if (!last) dh += dh_rec.p[ro];
        const float tc = tanhf(c.p[view_off(c, n, y, x, f)]);
        const size_t co = view_off(dc_next, n, y, x, f);
        float dc = dh * o * (1.f - tc * tc);
        if (!last) dc += dc_next.p[co];
        const float cprev = first ? 0.f : cp.p[view_off(cp, n, y, x, f)];
        const size_t dzo = view_off(dz, n, y, x, 0);
        dz.p[dzo + f] = dc * g * dhsig(zi);
        dz.p[dzo + F + f] = dc * cprev * dhsig(zf);
        dz.p[dzo + 2 * F + f] = dc * i * (1.f - g * g);
        dz.p[dzo + 3 * F + f] = dh * tc * dhsig(zq);
        dc_next.p[co] = dc * fg;
    }
}
inline int ew_blocks(size_t n) { return (int)std::max<size_t>(1, std::min<size_t>(cdivz(n, 256), 8192)); }
}  // namespace

void convlstm_gates_forward(hipStream_t s, const TView& z, const TView& c_prev, const TView& c, const TView& h,
                            const TView& out, int relu, int first) {
    const size_t total = (size_t)c.N * c.H * c.W * c.C;
    ProfScope ps(s, "convlstm_gates_fwd", 0.0, 4.0 * (double)total * 8);
    DL4DS_LAUNCH(gates_fwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, z, c_prev, c, h, out, relu, first, total);
    HIP_CHECK(hipGetLastError());
}
void convlstm_gates_backward(hipStream_t s, const TView& z, const TView& c_prev, const TView& c, const TView& out,
                             const TView& dout, const TView& dh_rec, const TView& dc_next, const TView& dz, int relu,
                             int first, int last) {
    const size_t total = (size_t)c.N * c.H * c.W * c.C;
    ProfScope ps(s, "convlstm_gates_bwd", 0.0, 4.0 * (double)total * 15);
    DL4DS_LAUNCH(gates_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, z, c_prev, c, out, dout, dh_rec, dc_next,
                       dz, relu, first, last, total);
    HIP_CHECK(hipGetLastError());
}
