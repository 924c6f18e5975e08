// One conditional-GAN optimisation step -- dl4ds/training/cgan.py:575-639 (train_step) with generator_loss
// (:525-553, total = BCE(1, D(fake)) + lambda * px_loss) and discriminator_loss (:556-572).
//
// The reference runs D twice (real, fake) under two gradient tapes.  Here D runs ONCE on a 2B batch
// [real ; fake] (same weights, per-sample independent), then two backward passes through it:
//   pass 1: dL_D/dp on all 2B rows  -> D parameter gradients (no input gradients)
//   pass 2: dL_G,gan/dp on the fake half only -> gradient w.r.t. D's HR input = dL/d(generated), no D parameter grads
// and the generator back-propagates lambda * dpx/dgen + (pass-2 result).
// Round 3: the conditioning branch is evaluated once for both halves (Graph::plan_shared) and pass 2 is replaced by a per-sample
// rescaling of pass 1's gradients at D's HR input layer (cgan_ratio_kernel below) where D has no batch statistics.
#include "graph.h"
#include "runtime.h"
#include "dist.h"
#include "prof.h"
#include <cmath>

// Pass 2 without a second backward pass (round 3).  D is per-sample independent (no batch statistics), its output is one
// probability p_i per sample, and back-propagation is linear in the output gradient: for a fake sample i every gradient pass 1
// left inside D is u1_i * (dp_i / d.) with u1_i = dBCE(p_i, 0)/dp, and pass 2 wants u2_i * (dp_i / d.) with u2_i = dBCE(p_i, 1)/dp.
// Both are zero exactly where Keras' clip(p, eps, 1 - eps) cuts the gradient, elsewhere u2_i / u1_i = -(1 - p_i) / p_i.  So the
// output gradient of the op(s) that read D's HR input is scaled per sample by that ratio (0 on the real half) and ONLY those
// ops back-propagate again (one 8 -> 1 dgrad in residual_discriminator) -- the rest of pass 2 (the HR branch's and the merge
// block's dgrads for the fake half) is gone.
__global__ void cgan_ratio_kernel(const float* __restrict__ p_fake, float* __restrict__ r, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * B) return;
    float v = 0.f;
    if (i >= B) {
        const float eps = 1e-7f, pr = p_fake[i - B];
        if (pr >= eps && pr <= 1.f - eps) { const float pc = fminf(fmaxf(pr, eps), 1.f - eps); v = -(1.f - pc) / pc; }
    }
    r[i] = v;
}
// v[image n, ...] *= r[n / nmul]  (the frames of one sample share its factor)
__global__ void scale_samples_kernel(TView v, const float* __restrict__ r, size_t per_img, int nmul, size_t total) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(e / per_img);
        size_t q = e - (size_t)n * per_img;
        const int c = (int)(q % v.C); q /= v.C;
        const int x = (int)(q % v.W);
        const int y = (int)(q / v.W);
        v.p[view_off(v, n, y, x, c)] *= r[n / nmul];
    }
}

Trainer* trainer_create(Graph* g, int loss_kind, const AdamCfg& cfg);
void graph_load_inputs(Graph& g, const float* const* inputs, int n_inputs, int B, bool is_host);
void trainer_apply_adam(Trainer& t);

struct CganTrainer {
    Trainer* G = nullptr;
    Trainer* D = nullptr;
    int px_kind = LOSS_MAE;
    float lam = 100.f;
    float* d_losses = nullptr;     // [0] gen_gan [1] gen_px [2] d_real [3] d_fake
    float* hr = nullptr;
    size_t hr_floats = 0;
    float* loss_ws = nullptr;
    size_t loss_ws_bytes = 0;
    int shared_plan = -1;          // discriminator: conditioning branch evaluated once for [real ; fake] (-1: not decided yet)
    int ratio_plan = -1;           // pass 2 by per-sample rescaling of pass 1's gradients (-1: not decided yet)
    std::vector<int> ratio_ops;    // ops of D that read its gradient-taking input
    float* ratio = nullptr;        // [2B] per-sample factors
    int ratio_n = 0;
};

CganTrainer* cgan_create(Graph* gen, Graph* disc, int px_loss_kind, float lr, float beta1, float lam) {
    DL4DS_REQUIRE(disc->inputs.size() == 2, "discriminator must have two inputs (lr/conditioning, hr/generated)");
    DL4DS_REQUIRE(disc->tensors[disc->inputs[1]].requires_grad, "discriminator HR input must be created with requires_grad");
    const GTensor& go = gen->tensors[gen->outputs.at(0)];
    const GTensor& dr = disc->tensors[disc->inputs[1]];
    DL4DS_REQUIRE(go.H == dr.H && go.W == dr.W && go.C == dr.C && go.nmul == dr.nmul,
                  "generator output and discriminator HR input shapes differ");
    AdamCfg c;
    c.lr0 = c.lr1 = lr;
    c.beta1 = beta1;
    CganTrainer* t = new CganTrainer();
    t->G = trainer_create(gen, px_loss_kind, c);
    t->D = trainer_create(disc, LOSS_MAE, c);
    t->px_kind = px_loss_kind;
    t->lam = lam;
    // (BatchNormalization inside the discriminator -- discriminator.py:38,50,70 -- keeps the statistics of the reference's two
    //  calls apart inside the merged [real ; fake] batch: two statistics groups, switched on for the duration of cgan_step only)
    HIP_CHECK(hipMalloc((void**)&t->d_losses, 8 * sizeof(float)));
    HIP_CHECK(hipMemset(t->d_losses, 0, 8 * sizeof(float)));
    return t;
}

void cgan_destroy(CganTrainer* t) {
    if (!t) return;
    delete t->G;
    delete t->D;
    if (t->d_losses) (void)hipFree(t->d_losses);
    if (t->hr) (void)hipFree(t->hr);
    if (t->loss_ws) (void)hipFree(t->loss_ws);
    if (t->ratio) (void)hipFree(t->ratio);
    delete t;
}
Trainer* cgan_disc_trainer(CganTrainer* t) { return t->D; }
Trainer* cgan_gen_trainer(CganTrainer* t) { return t->G; }
// genlr, dislr = learning_rates (cgan.py:271-278): one Adam per model, each with its own rate
void cgan_set_learning_rates(CganTrainer* t, float gen_lr, float disc_lr) {
    t->G->cfg.lr0 = t->G->cfg.lr1 = gen_lr;
    t->D->cfg.lr0 = t->D->cfg.lr1 = disc_lr;
}

void cgan_step(CganTrainer& t, const float* const* gen_inputs, int n_gen_inputs, const float* hr, int B, bool is_host,
               const float* dropout_keep_host, bool apply_update, float* losses_host) {
    Graph& G = *t.G->g;
    Graph& D = *t.D->g;
    hipStream_t s = G.stream;
    if (apply_update) dist_require_ready("dl4ds_cgan_step");   // WORLD_SIZE > 1 without a communicator is an error
    const GTensor& go = G.tensors[G.outputs[0]];
    const size_t hr_n = go.per_sample() * B;
    // ---- generator forward
    graph_load_inputs(G, gen_inputs, n_gen_inputs, B, is_host);
    if (hr_n > t.hr_floats) {
        HIP_CHECK(hipStreamSynchronize(s));
        if (t.hr) HIP_CHECK(hipFree(t.hr));
        HIP_CHECK(hipMalloc((void**)&t.hr, hr_n * sizeof(float)));
        t.hr_floats = hr_n;
    }
    const size_t lws = loss_workspace_bytes(t.px_kind, B * go.nmul, go.H, go.W, go.C);
    if (lws > t.loss_ws_bytes) {
        HIP_CHECK(hipStreamSynchronize(s));
        if (t.loss_ws) HIP_CHECK(hipFree(t.loss_ws));
        HIP_CHECK(hipMalloc((void**)&t.loss_ws, lws));
        t.loss_ws_bytes = lws;
    }
    HIP_CHECK(hipMemcpyAsync(t.hr, hr, hr_n * sizeof(float), is_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, s));
    G.forward(B, true);
    // ---- discriminator forward on [real ; fake]
    D.prepare(2 * B);
    GTensor& dlr = D.tensors[D.inputs[0]];
    GTensor& dhr = D.tensors[D.inputs[1]];
    const GTensor& glr = G.tensors[G.inputs[0]];
    DL4DS_REQUIRE(glr.per_sample() == dlr.per_sample(), "discriminator conditioning input must match the generator's first input");
    const size_t lr_n = dlr.per_sample() * B;
    HIP_CHECK(hipMemcpyAsync(dlr.data, glr.data, lr_n * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIP_CHECK(hipMemcpyAsync(dlr.data + lr_n, glr.data, lr_n * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIP_CHECK(hipMemcpyAsync(dhr.data, t.hr, hr_n * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIP_CHECK(hipMemcpyAsync(dhr.data + hr_n, go.data, hr_n * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (dropout_keep_host) {
        DL4DS_REQUIRE(!D.dropout_ops.empty(), "dropout mask given but the discriminator has no dropout op");
        GOp* dop = D.dropout_ops[0];
        size_t n = 0;
        for (size_t i = 0; i < D.ops.size(); ++i)
            if (D.ops[i].get() == dop) n = dop->saved_floats_per_sample(D) * 2 * B;
        dop->set_mask(D, dropout_keep_host, n);
    }
    // the conditioning branch sees the same array in both halves: evaluated once (Graph::plan_shared; decided per graph, once)
    if (t.shared_plan < 0) t.shared_plan = D.plan_shared({D.inputs[0]}) ? 1 : 0;
    // (restored on EVERY way out of this function: an exception between here and the end -- API_BEGIN / API_END turns it into a
    //  status -- must not leave the discriminator graph in shared mode for a later plain forward of the same model)
    struct SharedGuard {
        Graph& g;
        ~SharedGuard() {
            g.shared_groups = 1;
            for (auto& op : g.ops) op->set_batch_groups(1);
        }
    } shared_guard{D};
    for (auto& op : D.ops) op->set_batch_groups(2);          // BatchNormalization: [real ; fake] = two statistics groups
    D.shared_groups = t.shared_plan == 1 ? 2 : 1;
    D.forward(2 * B, true);
    GTensor& dout = D.tensors[D.outputs[0]];
    DL4DS_REQUIRE(dout.per_sample() == 1, "discriminator must output one probability per sample");
    float* p_real = dout.data;
    float* p_fake = dout.data + B;
    // ---- pass 1: discriminator loss -> D parameter gradients
    D.zero_grad_flags();
    bce_forward_backward(s, p_real, 1.f, B, 1.f, t.d_losses + 2, dout.grad, 0);
    bce_forward_backward(s, p_fake, 0.f, B, 1.f, t.d_losses + 3, dout.grad + B, 0);
    BwdCtx c1{2 * B, 0, 2 * B, true, false};
    D.grad_ready = nullptr;
    D.backward(c1);
    // data parallel: the discriminator's gradients are final here (the generator pass below only propagates to D's
    // inputs), so their all-reduce runs on the communication stream underneath pass 2 and the generator's backward
    if (apply_update) dist_allreduce_bucket_async(D.G, D.n_params, s, nullptr);
    // ---- pass 2: generator's adversarial loss through D (fake half, inputs only)
    bce_forward_backward(s, p_fake, 1.f, B, 1.f, t.d_losses + 0, dout.grad + B, 0);
    if (t.ratio_plan < 0) {
        // eligible: no normalisation ops (BatchNormalization couples the samples of a group), every reader of the HR input known
        bool ok = test_env("DL4DS_NO_CGAN_RATIO") == nullptr;
        t.ratio_ops.clear();
        for (size_t i = 0; i < D.ops.size() && ok; ++i) {
            GOp* op = D.ops[i].get();
            if (std::string(op->kind) == "norm") ok = false;
            bool reads = false;
            for (int tid : op->in_tids) reads = reads || (tid >= 0 && D.tensors[tid].is_input && D.tensors[tid].requires_grad);
            if (reads) { ok = ok && op->out_tid >= 0; t.ratio_ops.push_back((int)i); }
        }
        t.ratio_plan = (ok && !t.ratio_ops.empty()) ? 1 : 0;
    }
    bool partial = true;
    for (auto& op : D.ops) partial = partial && op->partial_batch_ok();
    if (t.ratio_plan == 1) {
        if (t.ratio_n < 2 * B) {
            HIP_CHECK(hipStreamSynchronize(s));
            if (t.ratio) HIP_CHECK(hipFree(t.ratio));
            HIP_CHECK(hipMalloc((void**)&t.ratio, (size_t)2 * B * sizeof(float)));
            t.ratio_n = 2 * B;
        }
        DL4DS_LAUNCH(cgan_ratio_kernel, dim3((2 * B + 255) / 256), dim3(256), 0, s, p_fake, t.ratio, B);
        HIP_CHECK(hipGetLastError());
        BwdCtx cr{2 * B, 0, 2 * B, false, true};
        for (int k = (int)t.ratio_ops.size() - 1; k >= 0; --k) {
            GOp* op = D.ops[t.ratio_ops[k]].get();
            const GTensor& to = D.tensors[op->out_tid];
            if (!to.grad_written) continue;
            TView gv = D.view(op->out_tid, 2 * B, true);
            const size_t per_img = (size_t)gv.H * gv.W * gv.C, total = per_img * gv.N;
            DL4DS_LAUNCH(scale_samples_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 8192)), dim3(256), 0, s, gv, t.ratio,
                               per_img, to.nmul, total);
            HIP_CHECK(hipGetLastError());
            op->backward(D, cr);
        }
    } else if (partial) {
        BwdCtx c2{2 * B, B, B, false, true};
        D.backward(c2);
    } else {
        // ConvLSTM2D / channel attention back-propagate whole batches only: run the full batch with a zero output
        // gradient on the real half (samples are independent in D, so the fake half's input gradient is unchanged)
        fill(s, dout.grad, (size_t)B, 0.f);
        BwdCtx c2{2 * B, 0, 2 * B, false, true};
        D.backward(c2);
    }
    // ---- generator backward: lambda * dpx/dgen + dgan/dgen
    G.zero_grad_flags();
    loss_forward_backward(s, t.px_kind, t.hr, go.data, go.grad, B * go.nmul, go.H, go.W, go.C, t.lam, t.d_losses + 1, 0,
                          t.loss_ws, t.loss_ws_bytes);
    TView src = make_view(dhr.grad + hr_n, B * go.nmul, go.H, go.W, go.C);
    TView dst = make_view(go.grad, B * go.nmul, go.H, go.W, go.C);
    view_axpy(s, src, dst, 1.f, 1);
    BwdCtx cg{B, 0, B, true, false};
    // the generator's gradient buckets follow as its backward pass completes them (Graph::plan_buckets)
    if (apply_update && dist_active()) {
        G.grad_ready = [](void*, float* grads, size_t n, hipStream_t st, hipStream_t aux) {
            dist_allreduce_bucket_async(grads, n, st, aux);
        };
    } else {
        G.grad_ready = nullptr;
    }
    G.backward(cg);
    // ---- data-parallel average (1/world folded into Adam) + the two Adam updates (cgan.py:608-617)
    if (apply_update) {
        dist_allreduce_wait(s);
        trainer_apply_adam(*t.G);
        trainer_apply_adam(*t.D);
    }
    if (losses_host) {
        float h[4];
        HIP_CHECK(hipMemcpyAsync(h, t.d_losses, 4 * sizeof(float), hipMemcpyDeviceToHost, s));
        dist_stream_sync(s, "dl4ds_cgan_step (loss read-back)");
        const float px = h[1] / t.lam;               // loss kernel was scaled by lambda
        losses_host[0] = h[0] + t.lam * px;          // gen_total
        losses_host[1] = h[0];                       // gen_gan
        losses_host[2] = px;                         // gen_px
        losses_host[3] = h[2] + h[3];                // disc
    }
}
