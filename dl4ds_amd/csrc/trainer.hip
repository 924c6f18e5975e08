// Per-step training arithmetic: forward -> loss(+grad) -> backward -> [RCCL gradient all-reduce] -> Keras Adam.
// Replaces the Keras `fit` inner step configured by SupervisedTrainer.run (dl4ds/training/supervised.py:336-353,
// 396-406) and train_step of the CGAN trainer (dl4ds/training/cgan.py:575-639).
#include "graph.h"
#include "runtime.h"
#include "dist.h"
#include <cmath>
#include <vector>

Trainer::~Trainer() {
    if (m) (void)hipFree(m);
    if (v) (void)hipFree(v);
    if (d_loss) (void)hipFree(d_loss);
    if (loss_ws) (void)hipFree(loss_ws);
    if (y_true) (void)hipFree(y_true);
}

Trainer* trainer_create(Graph* g, int loss_kind, const AdamCfg& cfg) {
    DL4DS_REQUIRE(g->finalized, "trainer: graph not finalized");
    DL4DS_REQUIRE(g->outputs.size() >= 1, "trainer: graph has no output");
    Trainer* t = new Trainer();
    t->g = g;
    t->cfg = cfg;
    t->loss_kind = loss_kind;
    const size_t bytes = std::max<size_t>(g->n_params, 4) * sizeof(float);
    HIP_CHECK(hipMalloc((void**)&t->m, bytes));
    HIP_CHECK(hipMalloc((void**)&t->v, bytes));
    HIP_CHECK(hipMemset(t->m, 0, bytes));
    HIP_CHECK(hipMemset(t->v, 0, bytes));
    HIP_CHECK(hipMalloc((void**)&t->d_loss, 8 * sizeof(float)));
    HIP_CHECK(hipMemset(t->d_loss, 0, 8 * sizeof(float)));
    return t;
}

void graph_load_inputs(Graph& g, const float* const* inputs, int n_inputs, int B, bool is_host) {
    DL4DS_REQUIRE(n_inputs == (int)g.inputs.size(), "wrong number of model inputs");
    g.prepare(B);
    for (int i = 0; i < n_inputs; ++i) {
        GTensor& t = g.tensors[g.inputs[i]];
        const size_t bytes = t.per_sample() * B * sizeof(float);
        if (inputs[i] == t.data) continue;       // already in place
        HIP_CHECK(hipMemcpyAsync(t.data, inputs[i], bytes, is_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice,
                                 g.stream));
    }
}

static void ensure_loss_buffers(Trainer& t, int B) {
    Graph& g = *t.g;
    const GTensor& o = g.tensors[g.outputs[0]];
    const size_t need = o.per_sample() * B;
    if (need > t.y_true_floats) {
        HIP_CHECK(hipStreamSynchronize(g.stream));
        if (t.y_true) HIP_CHECK(hipFree(t.y_true));
        HIP_CHECK(hipMalloc((void**)&t.y_true, need * sizeof(float)));
        t.y_true_floats = need;
    }
    const size_t ws = loss_workspace_bytes(t.loss_kind, B * o.nmul, o.H, o.W, o.C);
    if (ws > t.loss_ws_bytes) {
        HIP_CHECK(hipStreamSynchronize(g.stream));
        if (t.loss_ws) HIP_CHECK(hipFree(t.loss_ws));
        HIP_CHECK(hipMalloc((void**)&t.loss_ws, ws));
        t.loss_ws_bytes = ws;
    }
}

// forward + loss + backward; gradients land in g->G (un-averaged, this rank only)
void trainer_loss_and_grads(Trainer& t, const float* const* inputs, int n_inputs, const float* y_true, int B,
                            bool is_host, bool reduce_across_ranks) {
    Graph& g = *t.g;
    graph_load_inputs(g, inputs, n_inputs, B, is_host);
    ensure_loss_buffers(t, B);
    const GTensor& o = g.tensors[g.outputs[0]];
    const float* yt = y_true;
    if (is_host) {
        HIP_CHECK(hipMemcpyAsync(t.y_true, y_true, o.per_sample() * B * sizeof(float), hipMemcpyHostToDevice, g.stream));
        yt = t.y_true;
    }
    g.forward(B, true);
    g.zero_grad_flags();
    loss_forward_backward(g.stream, t.loss_kind, yt, o.data, o.grad, B * o.nmul, o.H, o.W, o.C, 1.f, t.d_loss, 0,
                          t.loss_ws, t.loss_ws_bytes);
    // (inputs created with dl4ds_graph_input_requires_grad get their gradient too: dl4ds_graph_tensor_ptr(id, grad = 1))
    bool input_grads = false;
    for (int i : g.inputs) input_grads = input_grads || g.tensors[i].requires_grad;
    BwdCtx c{B, 0, B, true, input_grads};
    // data parallel: buckets of the gradient arena are all-reduced on the communication stream as the backward pass
    // finishes them; trainer_step waits for the last one before Adam
    if (reduce_across_ranks && dist_active()) {
        g.grad_ready = [](void*, float* grads, size_t n, hipStream_t st, hipStream_t aux) {
            dist_allreduce_bucket_async(grads, n, st, aux);
        };
    } else {
        g.grad_ready = nullptr;
    }
    g.backward(c);
}

// model.evaluate: inference-mode forward (dropout off unless MC, BatchNormalization on its moving statistics, which are
// left untouched) + the loss value; no backward pass
void trainer_evaluate(Trainer& t, const float* const* inputs, int n_inputs, const float* y_true, int B, bool is_host) {
    Graph& g = *t.g;
    graph_load_inputs(g, inputs, n_inputs, B, is_host);
    ensure_loss_buffers(t, B);
    const GTensor& o = g.tensors[g.outputs[0]];
    const float* yt = y_true;
    if (is_host) {
        HIP_CHECK(hipMemcpyAsync(t.y_true, y_true, o.per_sample() * B * sizeof(float), hipMemcpyHostToDevice, g.stream));
        yt = t.y_true;
    }
    g.forward(B, false);
    // the fused loss kernels always emit dL/dpred; it lands in the output's gradient buffer and is never read
    loss_forward_backward(g.stream, t.loss_kind, yt, o.data, o.grad, B * o.nmul, o.H, o.W, o.C, 1.f, t.d_loss, 0,
                          t.loss_ws, t.loss_ws_bytes);
}

static float current_lr(const Trainer& t) {
    // PiecewiseConstantDecay: lr0 while iterations <= boundary, else lr1 (supervised.py:340-346)
    return ((double)t.step <= t.cfg.boundary) ? t.cfg.lr0 : t.cfg.lr1;
}

// The sticky device error word (runtime.h) ENDS the process's training: the optimiser kernel leaves parameters and moments untouched
// once it is set, every later host-side wait throws, and nothing clears it -- a caller that catches the exception and goes on would
// only advance the step counter here (and, in a multi-rank job, diverge from ranks that did not raise).  SupervisedEngine / CGANEngine
// do not catch it; the C header says the same (dl4ds_last_error: "device error word set").
void trainer_apply_adam(Trainer& t) {
    Graph& g = *t.g;
    const float lr = current_lr(t);
    t.step += 1;
    const double tt = (double)t.step;
    const float lr_t = (float)((double)lr * std::sqrt(1.0 - std::pow((double)t.cfg.beta2, tt)) /
                               (1.0 - std::pow((double)t.cfg.beta1, tt)));
    int rank = 0, world = 1;
    dist_world(rank, world);
    adam_update(g.stream, g.W, g.G, t.m, t.v, g.n_params, lr_t, t.cfg.beta1, t.cfg.beta2, t.cfg.eps,
                1.f / (float)world, device_error_word_if_any());
}

void trainer_step(Trainer& t, const float* const* inputs, int n_inputs, const float* y_true, int B, bool is_host,
                  float* loss_host) {
    Graph& g = *t.g;
    dist_require_ready("dl4ds_trainer_step");               // WORLD_SIZE > 1 without a communicator is an error, not a no-op
    trainer_loss_and_grads(t, inputs, n_inputs, y_true, B, is_host, true);
    dist_allreduce_wait(g.stream);                          // buckets were launched during the backward pass
    trainer_apply_adam(t);
    if (loss_host) {
        HIP_CHECK(hipMemcpyAsync(loss_host, t.d_loss, sizeof(float), hipMemcpyDeviceToHost, g.stream));
        dist_stream_sync(g.stream, "dl4ds_trainer_step (loss read-back)");
    }
}
