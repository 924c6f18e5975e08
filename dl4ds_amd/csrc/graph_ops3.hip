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

}  // namespace

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
