// Graph ops of the block variants: LayerNormalization / BatchNormalization with the fused
// activation that follows them in every dl4ds block.  See graph.h for the runtime contract, norm.hip for the kernels.
#include "graph.h"
#include "prof.h"
#include <algorithm>
#include <cstdlib>

namespace {

inline bool wants_grad(const Graph& g, int tid, const BwdCtx& c) {
    const GTensor& t = g.tensors[tid];
    return t.requires_grad && (!t.is_input || c.input_grads) && (c.param_grads || t.dep_grad_input || exp_env("DL4DS_NO_BWD_PRUNE") != nullptr);
}

// ============================================================================================ LayerNorm / BatchNorm
struct NormOp : GOp {
    int in, out, gamma, beta, mov_mean = -1, mov_var = -1;
    bool batch = false;
    float eps = 1e-3f;
    int relu = 0;
    int groups = 1;          // BatchNormalization: independent sub-batches of one forward batch (set_batch_groups)
    int fwd_groups = 1;      // ... as used by the last training-mode forward
    NormOp() { kind = "norm"; }
    size_t npix(Graph& g, int B) const { const GTensor& t = g.tensors[in]; return (size_t)B * t.nmul * t.H * t.W; }
    size_t workspace_bytes(Graph& g, int) override { return norm_workspace_bytes(g.tensors[in].C); }
    // batch statistics (mean, 1/std) of the last training forward; 2*C floats per group (a group holds >= 1 sample, so
    // 2*C per sample is always enough room)
    size_t saved_floats_per_sample(Graph& g) override { return batch ? 2 * (size_t)g.tensors[in].C : 0; }
    void set_batch_groups(int n) override { groups = std::max(n, 1); }
    void forward(Graph& g, int B, bool training) override {
        const GTensor& t = g.tensors[in];
        if (batch) {
            // one call per group: the reference's discriminator sees the real and the generated batch in two calls
            // (cgan.py:599-600), each with its own batch statistics and its own moving-average update, real first
            const int ng = (training && groups > 1 && B % groups == 0) ? groups : 1;
            if (training) fwd_groups = ng;
            const size_t gs = (size_t)(B / ng) * t.per_sample();
            for (int k = 0; k < ng; ++k)
                batchnorm_forward(g.stream, t.data + k * gs, g.wp(gamma), g.wp(beta), g.wp(mov_mean), g.wp(mov_var),
                                  g.tensors[out].data + k * gs, saved + (size_t)k * 2 * t.C, npix(g, B / ng), t.C, eps, 0.99f, training,
                                  relu, g.workspace, g.workspace_bytes);
        } else {
            layernorm_forward(g.stream, t.data, g.wp(gamma), g.wp(beta), g.tensors[out].data, npix(g, B), t.C, eps, relu);
        }
    }
    // a partial backward pass is defined for whole groups only (checked in backward)
    bool partial_batch_ok() const override { return !batch || groups > 1; }
    void backward(Graph& g, const BwdCtx& c) override {
        if (!g.tensors[out].grad_written) return;
        const GTensor& t = g.tensors[in];
        const int cnt = c.b_cnt < 0 ? c.B : c.b_cnt;
        const bool dx = wants_grad(g, in, c);
        const size_t off = (size_t)c.b_off * t.per_sample(), n = (size_t)cnt * t.nmul * t.H * t.W;
        float* dxp = dx ? t.grad + off : nullptr;
        float* dg = c.param_grads ? g.gp(gamma) : nullptr;
        float* db = c.param_grads ? g.gp(beta) : nullptr;
        const int accw = g.params[gamma].grad_written;
        if (batch) {
            // the statistics couple every sample of a group: back-propagation runs over whole groups
            const int ng = fwd_groups, per = c.B / ng;
            DL4DS_REQUIRE(c.B % ng == 0 && c.b_off % per == 0 && cnt % per == 0,
                          "batchnorm: backward over part of a statistics group is not defined");
            const size_t gs = (size_t)per * t.per_sample();
            for (int k = c.b_off / per, first = 1; k < (c.b_off + cnt) / per; ++k, first = 0)
                batchnorm_backward(g.stream, t.data + k * gs, g.tensors[out].data + k * gs, g.tensors[out].grad + k * gs, g.wp(gamma),
                                   saved + (size_t)k * 2 * t.C, dx ? t.grad + k * gs : nullptr, t.grad_written, dg, db,
                                   first ? accw : 1, (size_t)per * t.nmul * t.H * t.W, t.C, relu, g.workspace, g.workspace_bytes);
        } else {
            layernorm_backward(g.stream, t.data + off, g.tensors[out].data + off, g.tensors[out].grad + off, g.wp(gamma), dxp,
                               t.grad_written, dg, db, accw, n, t.C, eps, relu, g.workspace, g.workspace_bytes);
        }
        if (dx) g.tensors[in].grad_written = true;
        if (c.param_grads) g.params[gamma].grad_written = g.params[beta].grad_written = true;
    }
};

// ============================================================================================ Conv2D -> Conv2D 1x1, folded
// A KSxKS convolution (optionally storing through depth_to_space(r)) that is followed by a 1x1 convolution with nothing
// in between -- no activation, no other consumer -- is ONE linear map of the input:
//   W_eff[ky,kx,ci,(ij,co)] = sum_cm W1[ky,kx,ci,(ij,cm)] * W2[cm,co],   b_eff[(ij,co)] = sum_cm b1[(ij,cm)] * W2[cm,co] + b2[co]
// (ij = the r*r sub-pixel positions of depth_to_space; r = 1: plain).  This is the closing pair of every
// post-upsampling model without auxiliary / localized branches: SubpixelConvolutionBlock's last conv2x (48 -> 4 x 48
// at half the HR grid) or ResizeConvolutionBlock's conv (48 -> 48 at the HR grid) followed by TransitionBlock
// 'TransitionLast' (1x1, 48 -> 8)  (sp_postups.py:172-177,203; blocks.py:433-454,485-491,301-309).  Evaluating the
// composition needs Cm / Co times fewer multiply-adds on the largest grid of the model and never materialises the
// Cm-channel HR activation or its gradient.  The trainable variables stay W1, b1, W2, b2: W_eff / b_eff are rebuilt from
// them in every forward pass, and the backward pass unfolds the gradient of the effective filter by the chain rule
//   dW1[k,(ij,cm)] += sum_co dW_eff[k,(ij,co)] W2[cm,co]      dW2[cm,co] += sum_{k,ij} W1[k,(ij,cm)] dW_eff[k,(ij,co)]
//   db1[(ij,cm)]   += sum_co db_eff[(ij,co)] W2[cm,co]                      + sum_ij b1[(ij,cm)] db_eff[(ij,co)]
//   db2[co]        += sum_ij db_eff[(ij,co)]
// so losses, gradients and Adam updates equal the unfolded graph's up to fp32 rounding (tests/test_gpu_models.py runs
// both against the unfolded oracle).  DL4DS_NO_FOLD=1 makes the builders emit the two separate convolutions.
__global__ void fold_weights_kernel(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                    const float* __restrict__ b2, float* __restrict__ weff, float* __restrict__ beff, int K, int R2,
                                    int Cm, int Co) {
    const int ncol = R2 * Co;
    const int total = (K + 1) * ncol;                  // row K: the bias
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int col = e % ncol, k = e / ncol;
        const int ij = col / Co, co = col - ij * Co;
        const float* src = (k < K) ? w1 + (size_t)k * R2 * Cm + ij * Cm : b1 + ij * Cm;
        float a = 0.f;
        if (k < K || b1) for (int cm = 0; cm < Cm; ++cm) a = fmaf(src[cm], w2[cm * Co + co], a);
        if (k < K) weff[(size_t)k * ncol + col] = a;
        else beff[col] = a + (b2 ? b2[co] : 0.f);
    }
}
// dW1 / db1 part of the unfolding: one thread per (k | bias row, ij, cm)
__global__ void unfold_w1_kernel(const float* __restrict__ dweff, const float* __restrict__ dbeff, const float* __restrict__ w2,
                                 float* __restrict__ dw1, float* __restrict__ db1, int K, int R2, int Cm, int Co, int acc_w, int acc_b) {
    const int total = (K + 1) * R2 * Cm;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int cm = e % Cm;
        const int ij = (e / Cm) % R2;
        const int k = e / (Cm * R2);
        if (k == K && !db1) continue;
        const float* src = (k < K) ? dweff + (size_t)k * R2 * Co + ij * Co : dbeff + ij * Co;
        float a = 0.f;
        for (int co = 0; co < Co; ++co) a = fmaf(src[co], w2[cm * Co + co], a);
        float* d = (k < K) ? dw1 + (size_t)k * R2 * Cm + ij * Cm + cm : db1 + ij * Cm + cm;
        const int acc = (k < K) ? acc_w : acc_b;
        *d = acc ? *d + a : a;
    }
}
// dW2 / db2: one wavefront per (cm, co) (and one per co for db2), lanes stride over the K * R2 (+ R2 bias) terms
__global__ void __launch_bounds__(256) unfold_w2_kernel(const float* __restrict__ dweff, const float* __restrict__ dbeff,
                                                        const float* __restrict__ w1, const float* __restrict__ b1,
                                                        float* __restrict__ dw2, float* __restrict__ db2, int K, int R2, int Cm, int Co,
                                                        int acc_w, int acc_b) {
    const int lane = threadIdx.x & 63;
    const int nout = Cm * Co + Co;
    for (int o = blockIdx.x * 4 + (threadIdx.x >> 6); o < nout; o += gridDim.x * 4) {
        double a = 0.0;
        if (o < Cm * Co) {
            const int cm = o / Co, co = o - cm * Co;
            for (int t = lane; t < K * R2; t += 64) {
                const int k = t / R2, ij = t - k * R2;
                a += (double)w1[(size_t)k * R2 * Cm + ij * Cm + cm] * (double)dweff[(size_t)k * R2 * Co + ij * Co + co];
            }
            if (b1)
                for (int ij = lane; ij < R2; ij += 64) a += (double)b1[ij * Cm + cm] * (double)dbeff[ij * Co + co];
        } else if (db2) {
            const int co = o - Cm * Co;
            for (int ij = lane; ij < R2; ij += 64) a += (double)dbeff[ij * Co + co];
        }
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) a += __shfl_xor(a, m, 64);
        if (lane) continue;
        if (o < Cm * Co) dw2[o] = acc_w ? dw2[o] + (float)a : (float)a;
        else if (db2) { const int co = o - Cm * Co; db2[co] = acc_b ? db2[co] + (float)a : (float)a; }
    }
}

// Auxiliary-input form (sp_postups.py:184-203 with HR static variables): TransitionLast reads Concatenate([x_up, s]), and a 1x1
// convolution of a concatenation is the sum of two 1x1 convolutions, W2 = [W2x (Cm rows); W2s (Cs rows)]:
//   out = act( conv_eff(x)  +  conv1x1(s; W2s)  + b_eff ).
// The second term is an ordinary 1x1 convolution of the HR auxiliary tensor s into a private buffer, which the composed
// convolution adds in its epilogue (read through the same depth_to_space view it stores through); backward: d(s) and dW2s are
// the 1x1 dgrad / wgrad of the (masked) output gradient, dW2x / db2 / dW1 / db1 unfold as before.  Same variables as the
// unfolded graph: 'TransitionLast/conv/kernel' has Cm + Cs rows.
struct FoldedConvOp : GOp {
    int in, out, w1, b1, w2, b2, KS, r, Cm, Co, relu;
    int aux = -1, Cs = 0;          // HR auxiliary tensor and its channel count (rows Cm .. Cm+Cs of w2)
    size_t weff_off = 0, beff_off = 0, dweff_off = 0, wt_off = 0, wt_aux_off = 0;
    FoldedConvOp() { kind = "conv2d_folded"; }
    size_t saved_floats_per_sample(Graph& g) override { return aux >= 0 ? g.tensors[out].per_sample() : 0; }
    float* w2s(Graph& g) const { return g.wp(w2) + (size_t)Cm * Co; }
    float* gw2s(Graph& g) const { return g.gp(w2) + (size_t)Cm * Co; }
    TView ts_view(Graph& g, int B, int bo, int bc, bool through_d2s) {     // conv1x1(s; W2s): plain HR tensor / as the d2s operand
        const GTensor& ti = g.tensors[in];
        const GTensor& to = g.tensors[out];
        float* base = saved + (size_t)bo * to.per_sample();
        const int N = (bc < 0 ? B : bc) * to.nmul;
        if (through_d2s && r > 1) return make_view_d2s(base, N, ti.H, ti.W, ncol(), r);
        return make_view(base, N, to.H, to.W, Co);
    }
    int ncol() const { return r * r * Co; }
    int krows(Graph& g) const { return KS * KS * g.tensors[in].C; }
    void on_finalize(Graph& g) override {
        const size_t n = (size_t)krows(g) * ncol();
        weff_off = g.reserve_wt(n);
        beff_off = g.reserve_wt(ncol());
        dweff_off = g.reserve_wt(n + ncol());          // [dW_eff | db_eff]
        wt_off = g.reserve_wt(n);                      // dgrad arrangement of W_eff (W_eff itself is rebuilt in forward)
        g.add_wt_job(weff_off, true, wt_off, KS * KS, g.tensors[in].C, ncol());
        if (aux >= 0) {
            wt_aux_off = g.reserve_wt((size_t)Cs * Co);
            g.add_wt_job(g.params[w2].offset + (size_t)Cm * Co, false, wt_aux_off, 1, Cs, Co);
        }
        GTensor& t = g.tensors[out];
        bool is_output = false;
        for (int o : g.outputs) is_output |= (o == out);
        t.grad_masked = relu && !is_output && (t.n_conv_in + t.n_masking) >= 1 && t.n_add_in == 0 && t.n_other == 0 &&
                        !exp_env("DL4DS_NO_MASK_FUSION");
    }
    TView out_view(Graph& g, bool grad, int B, int bo, int bc) {
        const GTensor& ti = g.tensors[in];
        const GTensor& to = g.tensors[out];
        float* base = (grad ? to.grad : to.data) + (size_t)bo * to.per_sample();
        const int N = (bc < 0 ? B : bc) * to.nmul;
        if (r > 1) return make_view_d2s(base, N, ti.H, ti.W, ncol(), r);
        return make_view(base, N, ti.H, ti.W, ncol());
    }
    size_t workspace_bytes(Graph& g, int B) override {
        TView x = g.view(in, B, false);
        TView dz = make_view(nullptr, x.N, x.H, x.W, ncol());
        size_t ws = std::max(conv2d_wgrad_workspace_bytes(x, dz, KS), bias_grad_workspace_bytes(dz));
        if (aux >= 0) {
            TView s = g.view(aux, B, false);
            ws = std::max(ws, conv2d_wgrad_workspace_bytes(s, make_view(nullptr, s.N, s.H, s.W, Co), 1));
        }
        return ws;
    }
    void forward(Graph& g, int B, bool) override {
        const int K = krows(g);
        float* weff = g.Wt + weff_off;
        float* beff = g.Wt + beff_off;
        {
            ProfScope ps(g.stream, "fold_weights", 2.0 * (K + 1) * ncol() * Cm, 4.0 * ((double)K * r * r * Cm + (double)K * ncol()));
            const int total = (K + 1) * ncol();
            DL4DS_LAUNCH(fold_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, g.stream, g.wp(w1),
                               b1 >= 0 ? g.wp(b1) : nullptr, g.wp(w2), b2 >= 0 ? g.wp(b2) : nullptr, weff, beff, K, r * r, Cm, Co);
            HIP_CHECK(hipGetLastError());
        }
        ConvEpilogue ep;
        ep.bias = (b1 >= 0 || b2 >= 0) ? beff : nullptr;
        ep.relu = relu;
        if (aux >= 0) {
            ConvEpilogue none;
            conv2d_forward(g.stream, g.view(aux, B, false), w2s(g), 1, ts_view(g, B, 0, -1, false), none);
            ep.add = ts_view(g, B, 0, -1, true);
        }
        conv2d_forward(g.stream, g.view(in, B, false), weff, KS, out_view(g, false, B, 0, -1), ep);
    }
    void backward(Graph& g, const BwdCtx& c) override {
        if (!g.tensors[out].grad_written) return;
        const int K = krows(g), R2 = r * r;
        TView dY = out_view(g, true, c.B, c.b_off, c.b_cnt);
        if (relu && !g.tensors[out].grad_masked)
            bias_act_backward(g.stream, dY, out_view(g, false, c.B, c.b_off, c.b_cnt), dY, nullptr, 0, g.workspace, g.workspace_bytes);
        float* weff = g.Wt + weff_off;
        if (aux >= 0) {
            // the 1x1 convolution of the auxiliary tensor: its input and weight gradients from the same (masked) dZ, read as
            // the plain HR tensor it is
            const GTensor& to = g.tensors[out];
            const int cnt = c.b_cnt < 0 ? c.B : c.b_cnt;
            TView dZ = make_view(to.grad + (size_t)c.b_off * to.per_sample(), cnt * to.nmul, to.H, to.W, Co);
            if (c.param_grads)
                conv2d_wgrad(g.stream, g.view(aux, c.B, false, c.b_off, c.b_cnt), dZ, 1, gw2s(g), g.params[w2].grad_written, nullptr, 0,
                             g.workspace, g.workspace_bytes);
            if (wants_grad(g, aux, c)) {
                ConvEpilogue ep;
                ep.accumulate = g.tensors[aux].grad_written;
                if (g.tensors[aux].grad_masked) ep.mask = g.view(aux, c.B, false, c.b_off, c.b_cnt);
                conv2d_forward(g.stream, dZ, g.Wt + wt_aux_off, 1, g.view(aux, c.B, true, c.b_off, c.b_cnt), ep);
                g.tensors[aux].grad_written = true;
            }
        }
        if (c.param_grads) {
            float* dweff = g.Wt + dweff_off;
            float* dbeff = dweff + (size_t)K * ncol();
            conv2d_wgrad(g.stream, g.view(in, c.B, false, c.b_off, c.b_cnt), dY, KS, dweff, 0, dbeff, 0, g.workspace, g.workspace_bytes);
            ProfScope ps(g.stream, "unfold_weight_grads", 4.0 * K * ncol() * Cm, 4.0 * 3 * (double)K * R2 * Cm);
            const int n1 = (K + 1) * R2 * Cm;
            DL4DS_LAUNCH(unfold_w1_kernel, dim3((n1 + 255) / 256), dim3(256), 0, g.stream, dweff, dbeff, g.wp(w2), g.gp(w1),
                               b1 >= 0 ? g.gp(b1) : nullptr, K, R2, Cm, Co, (int)g.params[w1].grad_written,
                               b1 >= 0 ? (int)g.params[b1].grad_written : 0);
            const int n2 = Cm * Co + Co;
            DL4DS_LAUNCH(unfold_w2_kernel, dim3((n2 + 3) / 4), dim3(256), 0, g.stream, dweff, dbeff, g.wp(w1),
                               b1 >= 0 ? g.wp(b1) : nullptr, g.gp(w2), b2 >= 0 ? g.gp(b2) : nullptr, K, R2, Cm, Co,
                               (int)g.params[w2].grad_written, b2 >= 0 ? (int)g.params[b2].grad_written : 0);
            HIP_CHECK(hipGetLastError());
            g.params[w1].grad_written = g.params[w2].grad_written = true;
            if (b1 >= 0) g.params[b1].grad_written = true;
            if (b2 >= 0) g.params[b2].grad_written = true;
        }
        if (wants_grad(g, in, c)) {
            float* wt = g.Wt + wt_off;              // (Graph::refresh_dgrad_weights)
            ConvEpilogue ep;
            ep.accumulate = g.tensors[in].grad_written;
            if (g.tensors[in].grad_masked) ep.mask = g.view(in, c.B, false, c.b_off, c.b_cnt);
            conv2d_forward(g.stream, dY, wt, KS, g.view(in, c.B, true, c.b_off, c.b_cnt), ep);
            g.tensors[in].grad_written = true;
        }
    }
};

// ============================================================================================ strided slice (H, W)
// y[n, i, j, :] = x[n, oy + i*step, ox + j*step, :].  With the stride-1 'same' convolution in front of it this is
// Conv2D(strides=2) (discriminator.py:53-60): TF's 'same' padding for stride 2 puts the window of output i at rows
// 2i-pt .. 2i-pt+2 with pt = (H odd), i.e. the stride-1 output at row 2i + 1 - pt; 'valid' reads row 2i + 1.
// step == 1 is Cropping2D (discriminator.py:56).
__global__ void slice_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C, int Ho, int Wo, int oy,
                                 int ox, int step, size_t total) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        size_t r = e / C;
        const int j = (int)(r % Wo); r /= Wo;
        const int i = (int)(r % Ho);
        const size_t n = r / Ho;
        y[e] = x[((n * H + oy + i * step) * W + ox + j * step) * C + c];
    }
}
// dx (+)= scatter(dy): every input element looks up whether an output reads it
__global__ void slice_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int H, int W, int C, int Ho, int Wo, int oy,
                                 int ox, int step, size_t total, int accumulate) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        size_t r = e / C;
        const int xx = (int)(r % W); r /= W;
        const int yy = (int)(r % H);
        const size_t n = r / H;
        const int di = yy - oy, dj = xx - ox;
        float v = 0.f;
        if (di >= 0 && dj >= 0 && di % step == 0 && dj % step == 0 && di / step < Ho && dj / step < Wo)
            v = dy[((n * Ho + di / step) * Wo + dj / step) * C + c];
        dx[e] = accumulate ? dx[e] + v : v;
    }
}
// y (+)= x[:, :Ho, :Wo, :]  (crop of an (H, W) tensor; the gradient of the zero padding)
__global__ void slice_acc_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C, int Ho, int Wo, size_t total,
                                 int accumulate) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        size_t r = e / C;
        const int j = (int)(r % Wo); r /= Wo;
        const int i = (int)(r % Ho);
        const size_t n = r / Ho;
        const float v = x[((n * H + i) * W + j) * C + c];
        y[e] = accumulate ? y[e] + v : v;
    }
}
inline int ew_grid(size_t n) { return (int)std::max<size_t>(1, std::min<size_t>((n + 255) / 256, 8192)); }

struct SliceOp : GOp {
    int in, out, oy, ox, step;
    SliceOp() { kind = "slice"; }
    void forward(Graph& g, int B, bool) override {
        const GTensor& ti = g.tensors[in];
        const GTensor& to = g.tensors[out];
        const size_t total = to.per_sample() * B;
        DL4DS_LAUNCH(slice_fwd_kernel, dim3(ew_grid(total)), dim3(256), 0, g.stream, ti.data, to.data, ti.H, ti.W, ti.C,
                           to.H, to.W, oy, ox, step, total);
        HIP_CHECK(hipGetLastError());
    }
    void backward(Graph& g, const BwdCtx& c) override {
        if (!g.tensors[out].grad_written || !wants_grad(g, in, c)) return;
        const GTensor& ti = g.tensors[in];
        const GTensor& to = g.tensors[out];
        const int cnt = c.b_cnt < 0 ? c.B : c.b_cnt;
        const size_t total = ti.per_sample() * cnt;
        DL4DS_LAUNCH(slice_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, g.stream,
                           to.grad + (size_t)c.b_off * to.per_sample(), ti.grad + (size_t)c.b_off * ti.per_sample(), ti.H, ti.W,
                           ti.C, to.H, to.W, oy, ox, step, total, (int)ti.grad_written);
        HIP_CHECK(hipGetLastError());
        g.tensors[in].grad_written = true;
    }
};

// ============================================================================================ ZeroPadding2D (bottom / right)
// PadConcat (blocks.py:629-656) pads the smaller of two tensors with zeros at the bottom / right before concatenating:
// the adjoint pair of the slice above (forward = its scatter, backward = its gather, step 1, offset 0).
struct PadOp : GOp {
    int in, out;
    PadOp() { kind = "pad"; }
    void forward(Graph& g, int B, bool) override {
        const GTensor& ti = g.tensors[in];
        const GTensor& to = g.tensors[out];
        const size_t total = to.per_sample() * B;
        DL4DS_LAUNCH(slice_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, g.stream, ti.data, to.data, to.H, to.W, to.C, ti.H,
                           ti.W, 0, 0, 1, total, 0);
        HIP_CHECK(hipGetLastError());
    }
    void backward(Graph& g, const BwdCtx& c) override {
        if (!g.tensors[out].grad_written || !wants_grad(g, in, c)) return;
        const GTensor& ti = g.tensors[in];
        const GTensor& to = g.tensors[out];
        const int cnt = c.b_cnt < 0 ? c.B : c.b_cnt;
        const size_t total = ti.per_sample() * cnt;
        float* dx = ti.grad + (size_t)c.b_off * ti.per_sample();
        const float* dy = to.grad + (size_t)c.b_off * to.per_sample();
        DL4DS_LAUNCH(slice_acc_kernel, dim3(ew_grid(total)), dim3(256), 0, g.stream, dy, dx, to.H, to.W, to.C, ti.H, ti.W, total,
                           (int)ti.grad_written);
        HIP_CHECK(hipGetLastError());
        g.tensors[in].grad_written = true;
    }
};

// ============================================================================================ DepthwiseConv2D 7x7
struct DwConvOp : GOp {
    int in, out, w, b, KS;
    DwConvOp() { kind = "dwconv"; }
    size_t workspace_bytes(Graph& g, int) override { return dwconv_wgrad_workspace_bytes(g.tensors[in].C, KS); }
    void forward(Graph& g, int B, bool) override {
        const GTensor& t = g.tensors[in];
        dwconv_forward(g.stream, t.data, g.wp(w), b >= 0 ? g.wp(b) : nullptr, g.tensors[out].data, B * t.nmul, t.H, t.W, t.C, KS,
                       0, 0);
    }
    void backward(Graph& g, const BwdCtx& c) override {
        if (!g.tensors[out].grad_written) return;
        const GTensor& t = g.tensors[in];
        const int cnt = (c.b_cnt < 0 ? c.B : c.b_cnt) * t.nmul;
        const size_t off = (size_t)c.b_off * t.per_sample();
        const float* dy = g.tensors[out].grad + off;
        if (c.param_grads) {
            dwconv_wgrad(g.stream, t.data + off, dy, g.gp(w), b >= 0 ? g.gp(b) : nullptr, g.params[w].grad_written, cnt, t.H, t.W,
                         t.C, KS, g.workspace, g.workspace_bytes);
            g.params[w].grad_written = true;
            if (b >= 0) g.params[b].grad_written = true;
        }
        if (wants_grad(g, in, c)) {
            dwconv_forward(g.stream, dy, g.wp(w), nullptr, t.grad + off, cnt, t.H, t.W, t.C, KS, 1, t.grad_written);
            g.tensors[in].grad_written = true;
        }
    }
};

}  // namespace

int g_conv2d_folded(Graph& g, int in, int w1, int b1, int w2, int b2, int KS, int Cmid, int Cout, int relu, int d2s, int aux) {
    const GTensor ti = g.tensors.at(in);
    const int r = d2s > 1 ? d2s : 1;
    const int Cs = aux >= 0 ? g.tensors.at(aux).C : 0;
    DL4DS_REQUIRE(KS == 1 || KS == 3 || KS == 5 || KS == 7, "conv2d_folded: kernel size must be 1, 3, 5 or 7");
    DL4DS_REQUIRE(g.params.at(w1).n == (size_t)KS * KS * ti.C * r * r * Cmid, "conv2d_folded: first kernel size mismatch");
    DL4DS_REQUIRE(g.params.at(w2).n == (size_t)(Cmid + Cs) * Cout, "conv2d_folded: 1x1 kernel size mismatch");
    if (b1 >= 0) DL4DS_REQUIRE(g.params.at(b1).n == (size_t)r * r * Cmid, "conv2d_folded: first bias size mismatch");
    if (b2 >= 0) DL4DS_REQUIRE(g.params.at(b2).n == (size_t)Cout, "conv2d_folded: 1x1 bias size mismatch");
    if (aux >= 0) {
        const GTensor ta = g.tensors.at(aux);
        DL4DS_REQUIRE(ta.H == ti.H * r && ta.W == ti.W * r && ta.nmul == ti.nmul, "conv2d_folded: the auxiliary tensor must live on the output grid");
        DL4DS_REQUIRE((Cout & 3) == 0, "conv2d_folded: the auxiliary form needs Cout % 4 == 0");
    }
    const int out = g.add_tensor(ti.H * r, ti.W * r, Cout, ti.nmul, true, false);
    FoldedConvOp* op = new FoldedConvOp();
    g.ops.emplace_back(op);
    op->in = in; op->out = out; op->w1 = w1; op->b1 = b1; op->w2 = w2; op->b2 = b2; op->KS = KS; op->r = r; op->Cm = Cmid;
    op->Co = Cout; op->relu = relu; op->aux = aux; op->Cs = Cs;
    op->pids = {w1, b1, w2, b2};
    g.tensors[in].n_conv_in++;
    if (aux >= 0) g.tensors[aux].n_other++;
    op->out_tid = out; op->in_tids = {in, aux};
    g.tensors[out].dep_grad_input = g.tensors[in].dep_grad_input || (aux >= 0 && g.tensors[aux].dep_grad_input);
    return out;
}

int g_pad(Graph& g, int in, int Ho, int Wo) {
    const GTensor ti = g.tensors.at(in);
    DL4DS_REQUIRE(Ho >= ti.H && Wo >= ti.W, "pad: the padded grid must not be smaller than the input");
    const int out = g.add_tensor(Ho, Wo, ti.C, ti.nmul, true, false);
    PadOp* op = new PadOp();
    g.ops.emplace_back(op);
    op->in = in; op->out = out;
    g.tensors[in].n_other++;
    op->out_tid = out; op->in_tids = {in};
    g.tensors[out].dep_grad_input = g.tensors[in].dep_grad_input;
    return out;
}

int g_dwconv(Graph& g, int in, int w, int b, int KS) {
    const GTensor ti = g.tensors.at(in);
    DL4DS_REQUIRE(KS == 7, "depthwise conv: kernel size must be 7");
    DL4DS_REQUIRE(g.params.at(w).n == (size_t)KS * KS * ti.C, "depthwise conv: kernel size mismatch");
    DL4DS_REQUIRE(b < 0 || g.params.at(b).n == (size_t)ti.C, "depthwise conv: bias size mismatch");
    const int out = g.add_tensor(ti.H, ti.W, ti.C, ti.nmul, true, false);
    DwConvOp* op = new DwConvOp();
    g.ops.emplace_back(op);
    op->in = in; op->out = out; op->w = w; op->b = b; op->KS = KS;
    g.tensors[in].n_other++;
    op->pids = b >= 0 ? std::vector<int>{w, b} : std::vector<int>{w};
    op->out_tid = out; op->in_tids = {in};
    g.tensors[out].dep_grad_input = g.tensors[in].dep_grad_input;
    return out;
}

int g_slice(Graph& g, int in, int oy, int ox, int step, int Ho, int Wo) {
    const GTensor ti = g.tensors.at(in);
    DL4DS_REQUIRE(step >= 1 && oy >= 0 && ox >= 0 && Ho >= 1 && Wo >= 1, "slice: bad arguments");
    DL4DS_REQUIRE(oy + (Ho - 1) * step < ti.H && ox + (Wo - 1) * step < ti.W, "slice: window leaves the input grid");
    const int out = g.add_tensor(Ho, Wo, ti.C, ti.nmul, true, false);
    SliceOp* op = new SliceOp();
    g.ops.emplace_back(op);
    op->in = in; op->out = out; op->oy = oy; op->ox = ox; op->step = step;
    g.tensors[in].n_other++;
    op->out_tid = out; op->in_tids = {in};
    g.tensors[out].dep_grad_input = g.tensors[in].dep_grad_input;
    return out;
}

int g_norm(Graph& g, int in, int gamma, int beta, int mov_mean, int mov_var, int batch, float eps, int relu) {
    const GTensor ti = g.tensors.at(in);
    DL4DS_REQUIRE(g.params.at(gamma).n == (size_t)ti.C && g.params.at(beta).n == (size_t)ti.C, "norm: gamma/beta size mismatch");
    if (batch)
        DL4DS_REQUIRE(g.params.at(mov_mean).n == (size_t)ti.C && g.params.at(mov_var).n == (size_t)ti.C,
                      "norm: moving statistics size mismatch");
    DL4DS_REQUIRE(eps > 0.f, "norm: epsilon must be positive");
    const int out = g.add_tensor(ti.H, ti.W, ti.C, ti.nmul, true, false);
    NormOp* op = new NormOp();
    g.ops.emplace_back(op);
    op->in = in; op->out = out; op->gamma = gamma; op->beta = beta; op->batch = batch != 0; op->eps = eps; op->relu = relu;
    if (batch) { op->mov_mean = mov_mean; op->mov_var = mov_var; }
    g.tensors[in].n_other++;
    op->pids = batch ? std::vector<int>{gamma, beta, mov_mean, mov_var} : std::vector<int>{gamma, beta};
    op->out_tid = out; op->in_tids = {in};
    g.tensors[out].dep_grad_input = g.tensors[in].dep_grad_input;
    return out;
}
