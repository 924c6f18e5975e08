// Conv2DTranspose(k, strides=s, padding='same', use_bias=False) -- dl4ds/models/blocks.py:508-516 -- as a
// stride-1 implicit-GEMM convolution on the INPUT grid followed by a depth_to_space(s) store:
//
//   y[n, s*u+a, s*v+b, o] = sum_{dy,dx,c} x[n, u+dy, v+dx, c] * W[ky, kx, o, c],  ky = a + pb - s*dy, kx = b + pb - s*dx
//   (pb = (k-s)//2, taps with ky/kx outside [0,k) are zero)
//
// i.e. a (2R+1)x(2R+1) SAME cross-correlation with Cout' = s*s*Cout output channels ordered (a*s+b)*Cout + o --
// exactly the channel order depth_to_space expects.  For k=9, s=2: R=2 (5x5 virtual kernel, 81 of 100 taps
// non-zero).  Forward, dgrad and wgrad therefore reuse the MFMA kernels of conv.hip; only the weight
// re-arrangement (pack) and its adjoint (unpack of the weight gradient) are new.
#include "ops.h"
#include "prof.h"
#include <algorithm>

namespace {

struct DeconvGeom { int KS, s, pb, R, KV, Cin, Cout; };

DeconvGeom make_geom(int KS, int s, int Cin, int Cout) {
    DeconvGeom g;
    g.KS = KS; g.s = s; g.Cin = Cin; g.Cout = Cout;
    DL4DS_REQUIRE(KS >= s, "conv2d_transpose: kernel smaller than stride is not supported");
    g.pb = (KS - s) / 2;
    int R = 0;
    for (int a = 0; a < s; ++a)
        for (int ky = 0; ky < KS; ++ky) {
            const int num = a + g.pb - ky;
            if (num % s != 0) continue;
            R = std::max(R, std::abs(num / s));
        }
    g.R = R;
    g.KV = 2 * R + 1;
    DL4DS_REQUIRE(g.KV == 1 || g.KV == 3 || g.KV == 5, "conv2d_transpose: virtual kernel larger than 5x5");
    return g;
}

// wp[(ty*KV+tx)][c][(a*s+b)*Cout + o] = w[ky][kx][o][c] or 0
__global__ void deconv_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, DeconvGeom g) {
    const int CoutP = g.s * g.s * g.Cout;
    const size_t total = (size_t)g.KV * g.KV * g.Cin * CoutP;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int cop = (int)(e % CoutP);
        size_t r = e / CoutP;
        const int c = (int)(r % g.Cin);
        const int tap = (int)(r / g.Cin);
        const int ty = tap / g.KV, tx = tap % g.KV;
        const int o = cop % g.Cout, ab = cop / g.Cout;
        const int a = ab / g.s, b = ab % g.s;
        const int ky = a + g.pb - g.s * (ty - g.R), kx = b + g.pb - g.s * (tx - g.R);
        float v = 0.f;
        if (ky >= 0 && ky < g.KS && kx >= 0 && kx < g.KS) v = w[(((size_t)ky * g.KS + kx) * g.Cout + o) * g.Cin + c];
        wp[e] = v;
    }
}

// dw[ky][kx][o][c] (+)= dwp[tap(ky,kx)][c][(a*s+b)*Cout+o]  with a = (ky-pb) mod s, dy = (a+pb-ky)/s
__global__ void deconv_unpack_kernel(const float* __restrict__ dwp, float* __restrict__ dw, DeconvGeom g, int accumulate) {
    const int CoutP = g.s * g.s * g.Cout;
    const size_t total = (size_t)g.KS * g.KS * g.Cout * g.Cin;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % g.Cin);
        size_t r = e / g.Cin;
        const int o = (int)(r % g.Cout);
        r /= g.Cout;
        const int kx = (int)(r % g.KS), ky = (int)(r / g.KS);
        const int a = ((ky - g.pb) % g.s + g.s) % g.s, b = ((kx - g.pb) % g.s + g.s) % g.s;
        const int ty = (a + g.pb - ky) / g.s + g.R, tx = (b + g.pb - kx) / g.s + g.R;
        const float v = dwp[(((size_t)ty * g.KV + tx) * g.Cin + c) * CoutP + (a * g.s + b) * g.Cout + o];
        dw[e] = accumulate ? dw[e] + v : v;
    }
}

inline int ew_blocks(size_t n) { return (int)std::max<size_t>(1, std::min<size_t>(cdivz(n, 256), 4096)); }

struct Carve { float *wp, *wpt, *dwp, *rest; size_t rest_bytes; };
Carve carve(const DeconvGeom& g, float* ws, size_t ws_bytes) {
    const size_t nwp = ((size_t)g.KV * g.KV * g.Cin * g.s * g.s * g.Cout + 3) & ~(size_t)3;
    DL4DS_REQUIRE(ws_bytes >= 3 * nwp * sizeof(float), "conv2d_transpose: workspace too small");
    Carve c;
    c.wp = ws; c.wpt = ws + nwp; c.dwp = ws + 2 * nwp; c.rest = ws + 3 * nwp;
    c.rest_bytes = ws_bytes - 3 * nwp * sizeof(float);
    return c;
}

}  // namespace

size_t conv2d_transpose_workspace_bytes(const TView& in, const TView& out, int KS, int stride) {
    DeconvGeom g = make_geom(KS, stride, in.C, out.C);
    const size_t nwp = ((size_t)g.KV * g.KV * g.Cin * stride * stride * g.Cout + 3) & ~(size_t)3;
    TView dz = make_view(nullptr, in.N, in.H, in.W, stride * stride * out.C);
    return 3 * nwp * sizeof(float) + conv2d_wgrad_workspace_bytes(in, dz, g.KV);
}

void conv2d_transpose_forward(hipStream_t s, const TView& in, const float* w, int KS, int stride, const TView& out,
                              int relu, float* workspace, size_t workspace_bytes) {
    DL4DS_REQUIRE(out.H == in.H * stride && out.W == in.W * stride && out.N == in.N, "conv2d_transpose: output shape");
    DeconvGeom g = make_geom(KS, stride, in.C, out.C);
    Carve c = carve(g, workspace, workspace_bytes);
    const size_t total = (size_t)g.KV * g.KV * g.Cin * stride * stride * g.Cout;
    DL4DS_LAUNCH(deconv_pack_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, w, c.wp, g);
    HIP_CHECK(hipGetLastError());
    TView outv = make_view_d2s(out.p, in.N, in.H, in.W, stride * stride * out.C, stride);
    outv.nstride = out.nstride;
    outv.ld = out.ld;                 // (the output may be a channel slice of a Concatenate's buffer: pixel pitch > its channels)
    outv.vec = outv.vec && out.vec && (out.ld & 3) == 0;
    ConvEpilogue ep;
    ep.relu = relu;
    conv2d_forward(s, in, c.wp, g.KV, outv, ep);
}

void conv2d_transpose_dgrad(hipStream_t s, const TView& dz, const float* w, int KS, int stride, const TView& dx,
                            int accumulate, float* workspace, size_t workspace_bytes, const TView* relu_mask) {
    DeconvGeom g = make_geom(KS, stride, dx.C, dz.C);
    Carve c = carve(g, workspace, workspace_bytes);
    const int CoutP = stride * stride * dz.C;
    const size_t total = (size_t)g.KV * g.KV * g.Cin * CoutP;
    DL4DS_LAUNCH(deconv_pack_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, w, c.wp, g);
    HIP_CHECK(hipGetLastError());
    conv2d_dgrad_weights(s, c.wp, c.wpt, g.KV, g.Cin, CoutP);
    TView dzv = make_view_d2s(dz.p, dx.N, dx.H, dx.W, CoutP, stride);
    dzv.nstride = dz.nstride;
    dzv.ld = dz.ld;                   // (dz may be a channel slice of a Concatenate's gradient: pixel pitch > its channels)
    dzv.vec = dzv.vec && dz.vec && (dz.ld & 3) == 0;
    ConvEpilogue ep;
    ep.accumulate = accumulate;
    if (relu_mask && relu_mask->p) ep.mask = *relu_mask;
    conv2d_forward(s, dzv, c.wpt, g.KV, dx, ep);
}

void conv2d_transpose_wgrad(hipStream_t s, const TView& x, const TView& dz, int KS, int stride, float* dw, int accumulate,
                            float* workspace, size_t workspace_bytes) {
    DeconvGeom g = make_geom(KS, stride, x.C, dz.C);
    Carve c = carve(g, workspace, workspace_bytes);
    const int CoutP = stride * stride * dz.C;
    TView dzv = make_view_d2s(dz.p, x.N, x.H, x.W, CoutP, stride);
    dzv.nstride = dz.nstride;
    dzv.ld = dz.ld;
    dzv.vec = dzv.vec && dz.vec && (dz.ld & 3) == 0;
    conv2d_wgrad(s, x, dzv, g.KV, c.dwp, 0, nullptr, 0, c.rest, c.rest_bytes);
    const size_t total = (size_t)KS * KS * dz.C * x.C;
    DL4DS_LAUNCH(deconv_unpack_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, c.dwp, dw, g, accumulate);
    HIP_CHECK(hipGetLastError());
}
