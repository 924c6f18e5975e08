// Graph ops of the block variants: LayerNormalization / BatchNormalization with the fused
// activation that follows them in every dl4ds block.  See graph.h for the runtime contract, norm.hip for the kernels.
#include "graph.h"
#include <algorithm>

namespace {

inline bool wants_grad(const Graph& g, int tid, const BwdCtx& c) {
    const GTensor& t = g.tensors[tid];
    return t.requires_grad && (!t.is_input || c.input_grads);
}

// ============================================================================================ LayerNorm / BatchNorm
struct NormOp : GOp {
    int in, out, gamma, beta, mov_mean = -1, mov_var = -1;
    bool batch = false;
    float eps = 1e-3f;
    int relu = 0;
    NormOp() { kind = "norm"; }
    size_t npix(Graph& g, int B) const { const GTensor& t = g.tensors[in]; return (size_t)B * t.nmul * t.H * t.W; }
    size_t workspace_bytes(Graph& g, int) override { return norm_workspace_bytes(g.tensors[in].C); }
    // batch statistics (mean, 1/std) of the last training forward; 2*C floats in total, independent of B
    size_t saved_floats_per_sample(Graph& g) override { return batch ? 2 * (size_t)g.tensors[in].C : 0; }
    void forward(Graph& g, int B, bool training) override {
        const GTensor& t = g.tensors[in];
        if (batch)
            batchnorm_forward(g.stream, t.data, g.wp(gamma), g.wp(beta), g.wp(mov_mean), g.wp(mov_var), g.tensors[out].data,
                              saved, npix(g, B), t.C, eps, 0.99f, training, relu, g.workspace, g.workspace_bytes);
        else
            layernorm_forward(g.stream, t.data, g.wp(gamma), g.wp(beta), g.tensors[out].data, npix(g, B), t.C, eps, relu);
    }
    bool partial_batch_ok() const override { return !batch; }
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
            // the statistics couple every sample of the forward batch: a partial back-propagation has no meaning
            DL4DS_REQUIRE(c.b_off == 0 && cnt == c.B, "batchnorm: backward over part of the batch is not defined");
            batchnorm_backward(g.stream, t.data, g.tensors[out].data, g.tensors[out].grad, g.wp(gamma), saved, dxp,
                               t.grad_written, dg, db, accw, n, t.C, relu, g.workspace, g.workspace_bytes);
        } else {
            layernorm_backward(g.stream, t.data + off, g.tensors[out].data + off, g.tensors[out].grad + off, g.wp(gamma), dxp,
                               t.grad_written, dg, db, accw, n, t.C, eps, relu, g.workspace, g.workspace_bytes);
        }
        if (dx) g.tensors[in].grad_written = true;
        if (c.param_grads) g.params[gamma].grad_written = g.params[beta].grad_written = true;
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
        hipLaunchKernelGGL(slice_fwd_kernel, dim3(ew_grid(total)), dim3(256), 0, g.stream, ti.data, to.data, ti.H, ti.W, ti.C,
                           to.H, to.W, oy, ox, step, total);
        HIP_CHECK(hipGetLastError());
    }
    void backward(Graph& g, const BwdCtx& c) override {
        if (!g.tensors[out].grad_written || !wants_grad(g, in, c)) return;
        const GTensor& ti = g.tensors[in];
        const GTensor& to = g.tensors[out];
        const int cnt = c.b_cnt < 0 ? c.B : c.b_cnt;
        const size_t total = ti.per_sample() * cnt;
        hipLaunchKernelGGL(slice_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, g.stream,
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
        hipLaunchKernelGGL(slice_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, g.stream, ti.data, to.data, to.H, to.W, to.C, ti.H,
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
        hipLaunchKernelGGL(slice_acc_kernel, dim3(ew_grid(total)), dim3(256), 0, g.stream, dy, dx, to.H, to.W, to.C, ti.H, ti.W, total,
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

int g_pad(Graph& g, int in, int Ho, int Wo) {
    const GTensor ti = g.tensors.at(in);
    DL4DS_REQUIRE(Ho >= ti.H && Wo >= ti.W, "pad: the padded grid must not be smaller than the input");
    const int out = g.add_tensor(Ho, Wo, ti.C, ti.nmul, true, false);
    PadOp* op = new PadOp();
    g.ops.emplace_back(op);
    op->in = in; op->out = out;
    g.tensors[in].n_other++;
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
    return out;
}
