// Graph runtime + op implementations (forward and hand-written backward) -- see graph.h.
#include "graph.h"
#include <cstdlib>
#include "prof.h"
#include <algorithm>
#include <cstring>
#include <cmath>
#include <limits>

static void plan_concat_aliases(Graph& g);
static void plan_grad_aliases(Graph& g);

// ============================================================================================ Graph
Graph::~Graph() {
    if (W) wino_filters_release(W, W + n_params);
    if (Wt) wino_filters_release(Wt, Wt + wt_floats);
    for (float* p : allocations) (void)hipFree(p);
    if (own_arena) {
        if (W) (void)hipFree(W);
        if (G) (void)hipFree(G);
    }
    if (Wt) (void)hipFree(Wt);
    if (wt_jobs_dev) (void)hipFree(wt_jobs_dev);
    if (workspace) (void)hipFree(workspace);
    if (aux_workspace) (void)hipFree(aux_workspace);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
}

// per-launch profiling wants stand-alone kernel durations: no concurrency while the profiler is on
static bool aux_enabled(const Graph& g) { return g.aux_stream != nullptr && !prof().on; }

void Graph::fork_aux() {
    if (!aux_enabled(*this)) return;
    if (!ev_fork) {
        HIP_CHECK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
    }
    HIP_CHECK(hipEventRecord(ev_fork, stream));
    HIP_CHECK(hipStreamWaitEvent(aux_stream, ev_fork, 0));
    aux_used = true;
}

void Graph::join_aux() {
    if (!aux_stream || !aux_used) return;
    HIP_CHECK(hipEventRecord(ev_join, aux_stream));
    HIP_CHECK(hipStreamWaitEvent(stream, ev_join, 0));
    aux_used = false;
}

int Graph::add_tensor(int H, int W_, int C, int nmul, bool requires_grad, bool is_input) {
    DL4DS_REQUIRE(!finalized, "graph already finalized");
    DL4DS_REQUIRE(H > 0 && W_ > 0 && C > 0 && nmul > 0, "bad tensor shape");
    GTensor t;
    t.H = H; t.W = W_; t.C = C; t.nmul = nmul; t.requires_grad = requires_grad; t.is_input = is_input;
    tensors.push_back(t);
    if (is_input) inputs.push_back((int)tensors.size() - 1);
    return (int)tensors.size() - 1;
}

int Graph::add_param(size_t n) {
    DL4DS_REQUIRE(!finalized, "graph already finalized");
    GParam p;
    p.offset = n_params;
    p.n = n;
    n_params += (n + 3) & ~(size_t)3;          // keep every slice 16-byte aligned for float4 access
    params.push_back(p);
    return (int)params.size() - 1;
}

void Graph::finalize() {
    DL4DS_REQUIRE(!finalized, "graph already finalized");
    if (own_arena) {
        const size_t bytes = std::max<size_t>(n_params, 4) * sizeof(float);
        HIP_CHECK(hipMalloc((void**)&W, bytes));
        HIP_CHECK(hipMalloc((void**)&G, bytes));
        HIP_CHECK(hipMemset(W, 0, bytes));
        HIP_CHECK(hipMemset(G, 0, bytes));
    }
    for (auto& op : ops) op->on_finalize(*this);
    plan_concat_aliases(*this);
    plan_grad_aliases(*this);
    if (wt_floats) HIP_CHECK(hipMalloc((void**)&Wt, wt_floats * sizeof(float)));
    finalized = true;
    // the headline model's arena is 0.82 MB (latency-bound collective): a few buckets; U-Net (54 MB): ~4 MB each
    plan_buckets(std::max<size_t>(256 << 10, n_params * sizeof(float) / 12));
}

void Graph::plan_buckets(size_t target_bytes) {
    buckets.clear();
    if (params.empty()) return;
    // first forward op that reads each parameter (-1: unused -> final from the start)
    std::vector<int> first_use(params.size(), -1);
    for (int i = (int)ops.size() - 1; i >= 0; --i)
        for (int pid : ops[i]->pids)
            if (pid >= 0 && pid < (int)params.size()) first_use[pid] = i;
    GradBucket cur;
    cur.p_lo = 0; cur.off = params[0].offset; cur.ready_op = (int)ops.size();
    size_t end_prev = params[0].offset;
    for (int pid = 0; pid < (int)params.size(); ++pid) {
        cur.ready_op = std::min(cur.ready_op, first_use[pid]);
        const size_t end = (pid + 1 < (int)params.size()) ? params[pid + 1].offset : n_params;
        cur.p_hi = pid + 1;
        cur.n = end - cur.off;
        end_prev = end;
        // the bucket that holds the FIRST parameters of the arena is the last one to become final (its collective has only
        // the optimiser behind it to hide under): keep it small -- 256 KB is still latency-bound on xGMI
        const size_t want = buckets.empty() ? std::min<size_t>(target_bytes, 256 << 10) : target_bytes;
        if (cur.n * sizeof(float) >= want || pid + 1 == (int)params.size()) {
            buckets.push_back(cur);
            cur = GradBucket();
            cur.p_lo = pid + 1; cur.off = end; cur.ready_op = (int)ops.size();
        }
    }
    (void)end_prev;
}

void Graph::prepare(int B) {
    DL4DS_REQUIRE(finalized, "graph not finalized");
    DL4DS_REQUIRE(B > 0, "batch must be positive");
    if (B <= maxB) return;
    HIP_CHECK(hipStreamSynchronize(stream));
    for (float* p : allocations) HIP_CHECK(hipFree(p));
    allocations.clear();
    if (workspace) { HIP_CHECK(hipFree(workspace)); workspace = nullptr; }
    if (aux_workspace) { HIP_CHECK(hipFree(aux_workspace)); aux_workspace = nullptr; }
    // one slab for all activations + gradients + op-private saved buffers (288 GB HBM: no reuse games)
    size_t total = 0;
    auto bump = [&](size_t floats) { size_t o = total; total += (floats + 63) & ~(size_t)63; return o; };
    std::vector<size_t> doff(tensors.size()), goff(tensors.size()), soff(ops.size());
    for (size_t i = 0; i < tensors.size(); ++i) {
        doff[i] = tensors[i].alias_of >= 0 ? (size_t)-1 : bump(tensors[i].per_sample() * B);     // (aliased: inside another buffer)
        goff[i] = (tensors[i].requires_grad && !tensors[i].galias) ? bump(tensors[i].per_sample() * B) : (size_t)-1;
    }
    for (size_t i = 0; i < ops.size(); ++i) soff[i] = bump(ops[i]->saved_floats_per_sample(*this) * B + 64);
    float* slab = nullptr;
    HIP_CHECK(hipMalloc((void**)&slab, total * sizeof(float)));
    // once per (re)allocation: kernels that load 16-byte quads at dword alignment read floats that belong to a neighbouring
    // pixel / channel slice and multiply them by zero filter entries -- those floats must be finite from the first step on
    HIP_CHECK(hipMemsetAsync(slab, 0, total * sizeof(float), stream));
    allocations.push_back(slab);
    for (size_t i = 0; i < tensors.size(); ++i) {
        tensors[i].data = (doff[i] == (size_t)-1) ? nullptr : slab + doff[i];
        tensors[i].grad = (goff[i] == (size_t)-1) ? nullptr : slab + goff[i];
    }
    for (size_t i = tensors.size(); i-- > 0;) {               // a concatenation is created after its inputs: resolved first
        GTensor& t = tensors[i];
        if (t.alias_of >= 0) t.data = tensors[t.alias_of].data + t.alias_coff;
        if (t.galias) t.grad = tensors[t.alias_of].grad + t.alias_coff;
    }
    for (size_t i = 0; i < ops.size(); ++i) ops[i]->saved = slab + soff[i];
    size_t ws = 1 << 20;
    for (auto& op : ops) ws = std::max(ws, op->workspace_bytes(*this, B));
    workspace_bytes = ws;
    HIP_CHECK(hipMalloc((void**)&workspace, ws));
    if (aux_stream) HIP_CHECK(hipMalloc((void**)&aux_workspace, ws));
    maxB = B;
    for (auto& t : tensors) t.two_add_inplace = false;           // (re-planned below: ConvOp::plan_two_adds)
    for (auto& op : ops) op->on_prepare(*this);
    // gradient buffers shared between a fused add's output and its residual operand: resolved once more in REVERSE op order, so that a
    // chain (this op's output is a later op's shared / in-place residual, which re-pointed its gradient after this op had planned)
    // ends on the final buffer instead of a stale alias
    for (auto it = ops.rbegin(); it != ops.rend(); ++it) (*it)->on_resolve_aliases(*this);
}

TView Graph::view(int tid, int B, bool grad, int b_off, int b_cnt) const {
    const GTensor& t = tensors[tid];
    float* base = grad ? t.grad : t.data;
    DL4DS_REQUIRE(base != nullptr, "tensor buffer missing (no grad buffer / graph not prepared)");
    const int cnt = (b_cnt < 0) ? B : b_cnt;
    if ((!grad || t.galias) && t.alias_of >= 0) {
        const size_t img = (size_t)t.H * t.W * t.alias_ld;
        TView v = make_view(base + (size_t)b_off * t.nmul * img, cnt * t.nmul, t.H, t.W, t.C);
        v.ld = t.alias_ld;
        v.nstride = img;
        v.vec = v.vec && (t.alias_ld & 3) == 0;
        return v;
    }
    return make_view(base + (size_t)b_off * t.per_sample(), cnt * t.nmul, t.H, t.W, t.C);
}

bool Graph::plan_shared(const std::vector<int>& dup_inputs) {
    op_shared.assign(ops.size(), 0);
    t_boundary.assign(tensors.size(), 0);
    auto fail = [&]() { op_shared.clear(); t_boundary.clear(); return false; };
    if (test_env("DL4DS_NO_SHARED_BRANCH")) return fail();
    std::vector<char> tdup(tensors.size(), 0);
    for (int t : dup_inputs) {
        if (tensors.at(t).requires_grad) return fail();
        tdup[t] = 1;
    }
    bool any_shared = false;
    for (size_t i = 0; i < ops.size(); ++i) {
        GOp* op = ops[i].get();
        if (op->out_tid < 0) return fail();
        bool all = true, any = false;
        for (int t : op->in_tids) {
            if (t < 0) continue;
            any = true;
            all = all && tdup[t];
        }
        if (!(all && any)) continue;
        // per-sample ops without batch statistics, noise or cross-sample state only
        const std::string k = op->kind;
        const bool plain = k == "conv2d" || k == "add" || k == "act" || k == "concat" || k == "maxpool2" || k == "pad" || k == "slice";
        if (!plain || !op->partial_batch_ok()) return fail();
        op_shared[i] = 1;
        tdup[op->out_tid] = 1;
        any_shared = true;
    }
    if (!any_shared) return fail();
    for (int o : outputs) if (tdup[o]) return fail();
    for (size_t i = 0; i < ops.size(); ++i) {
        if (op_shared[i]) continue;
        for (int t : ops[i]->in_tids)
            if (t >= 0 && tdup[t] && !tensors[t].is_input) t_boundary[t] = 1;
    }
    // a tensor handed over to the rest of the graph must not be read inside the shared part as well (its gradient groups are
    // summed right before its producer's backward: a shared reader would have written one group only)
    for (size_t i = 0; i < ops.size(); ++i) {
        if (!op_shared[i]) continue;
        for (int t : ops[i]->in_tids)
            if (t >= 0 && t_boundary[t]) return fail();
    }
    return true;
}

void Graph::forward(int B, bool training) {
    prepare(B);
    wt_fresh = false;          // the filters may have been updated since the last backward pass
    // ... and so are the Winograd layers' transformed filters: all stale, those made from W re-made in one launch (conv_wino.hip)
    WinoPassGuard wino_pass(0);
    wino_filters_invalidate(W, W + n_params);
    if (Wt) wino_filters_invalidate(Wt, Wt + wt_floats);
    wino_filters_refresh(stream, W, W + n_params, 0);
    const bool sh = shared_groups > 1 && !op_shared.empty() && B % shared_groups == 0;
    for (size_t i = 0; i < ops.size(); ++i) {
        GOp* op = ops[i].get();
        if (sh && op_shared[i]) {
            const int Bh = B / shared_groups;
            op->forward(*this, Bh, training);
            if (t_boundary[op->out_tid])
                for (int gi = 1; gi < shared_groups; ++gi)
                    view_axpy(stream, view(op->out_tid, B, false, 0, Bh), view(op->out_tid, B, false, gi * Bh, Bh), 1.f, 0);
        } else {
            op->forward(*this, B, training);
        }
    }
}

void Graph::refresh_dgrad_weights() {
    if (wt_fresh || wt_jobs.empty()) { wt_fresh = true; return; }
    if (wt_jobs_dev == nullptr) {
        std::vector<DgradWeightsJob> host;
        int blocks = 0;
        for (const WtJob& j : wt_jobs) {
            DgradWeightsJob d;
            d.src = (j.src_in_wt ? Wt : W) + j.src_off;
            d.dst = Wt + j.dst_off;
            d.KK = j.KK; d.Cin = j.Cin; d.Cout = j.Cout; d.block0 = blocks;
            blocks += dgrad_weights_job_blocks(j.KK, j.Cin, j.Cout);
            host.push_back(d);
        }
        HIP_CHECK(hipMalloc(&wt_jobs_dev, host.size() * sizeof(DgradWeightsJob)));
        HIP_CHECK(hipMemcpyAsync(wt_jobs_dev, host.data(), host.size() * sizeof(DgradWeightsJob), hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipStreamSynchronize(stream));            // `host` goes out of scope
        wt_job_blocks = blocks;
    }
    conv2d_dgrad_weights_batched(stream, static_cast<const DgradWeightsJob*>(wt_jobs_dev), (int)wt_jobs.size(), wt_job_blocks);
    wt_fresh = true;
    wino_filters_refresh(stream, Wt, Wt + wt_floats, 1);       // the dgrad layers' transformed filters, from the arrangements just made
}

void Graph::zero_grad_flags() {
    for (auto& t : tensors) t.grad_written = false;
    for (auto& p : params) p.grad_written = false;
}

void Graph::backward(const BwdCtx& c) {
    WinoPassGuard wino_pass(1);
    refresh_dgrad_weights();
    for (auto& t : tensors) { t.grad_written = false; t.pending_add = nullptr; }
    for (int o : outputs) tensors[o].grad_written = true;      // seeded by the loss
    const bool bucketed = c.param_grads && grad_ready != nullptr && !buckets.empty();
    std::vector<char> sent(bucketed ? buckets.size() : 0, 0);
    auto flush_ready = [&](int done_op) {          // every op with index >= done_op has run its backward
        for (size_t k = 0; k < buckets.size(); ++k) {
            if (sent[k] || buckets[k].ready_op < done_op) continue;
            for (int pid = buckets[k].p_lo; pid < buckets[k].p_hi; ++pid)
                if (!params[pid].grad_written) { fill(stream, G + params[pid].offset, params[pid].n, 0.f); params[pid].grad_written = true; }
            // weight gradients of the bucket may still be running on the aux stream: the collective waits for it too
            grad_ready(grad_ready_ctx, G + buckets[k].off, buckets[k].n, stream, (aux_stream && aux_used) ? aux_stream : nullptr);
            sent[k] = 1;
        }
    };
    const bool sh = shared_groups > 1 && !op_shared.empty() && c.B % shared_groups == 0;
    const bool whole = c.b_off == 0 && (c.b_cnt < 0 || c.b_cnt == c.B);
    for (int i = (int)ops.size() - 1; i >= 0; --i) {
        if (sh && op_shared[i]) {
            // the shared part takes no input gradient (plan_shared): only a whole-batch pass with parameter gradients enters it
            if (whole && c.param_grads) {
                const int t = ops[i]->out_tid, Bh = c.B / shared_groups;
                if (t_boundary[t] && tensors[t].grad_written)
                    for (int gi = 1; gi < shared_groups; ++gi)
                        view_axpy(stream, view(t, c.B, true, gi * Bh, Bh), view(t, c.B, true, 0, Bh), 1.f, 1);
                BwdCtx ch = c;
                ch.b_off = 0; ch.b_cnt = Bh;
                ops[i]->backward(*this, ch);
            }
        } else {
            ops[i]->backward(*this, c);
        }
        if (bucketed) flush_ready(i);
    }
    join_aux();
    if (bucketed) flush_ready(-1);                   // buckets of parameters no op reads
    if (c.param_grads) {
        // parameters never reached by the backward pass get an explicit zero gradient
        for (auto& p : params)
            if (!p.grad_written) { fill(stream, G + p.offset, p.n, 0.f); p.grad_written = true; }
    }
}

namespace {

inline bool wants_grad(const Graph& g, int tid, const BwdCtx& c) {
    const GTensor& t = g.tensors[tid];
    // (a pass without parameter gradients -- the generator's adversarial gradient through the discriminator -- only needs the
    //  tensors that depend on an input that takes a gradient: the conditioning branch of the discriminator is skipped)
    return t.requires_grad && (!t.is_input || c.input_grads) && (c.param_grads || t.dep_grad_input || exp_env("DL4DS_NO_BWD_PRUNE") != nullptr);
}

// ChannelAttention2D between two convolutions (ConvBlock(attention=True) followed by another conv: blocks.py:87-103,
// sp_postups.py:204-211) costs three passes over the HR tensor when run as its own kernels.  Where the neighbouring
// convolutions run on kernels that can take over (decided in ChAttOp::on_prepare, per piece):
//   fuse_pool  : the producing conv emits the pooling partial sums from its epilogue       (no pooling pass)
//   fuse_scale : the consuming conv reads the attention INPUT through TView::sc = scale     (no scale pass, y never stored)
//   fuse_dx    : the producing conv's backward reads dY = att.out.grad * scale + dmean lazily (no dX pass)
// DL4DS_NO_TAIL_FUSION=1 keeps the three passes (A/B measurements, tests).
struct AttFusion {
    bool fuse_pool = false, fuse_scale = false, fuse_dx = false;
    float* pool = nullptr;      // [G][tiles][8]
    int pool_tiles = 0;
    float* scale = nullptr;     // [G][C]
    float* dmean = nullptr;     // [G][C]
    float* ds = nullptr;        // [G][C] sum over pixels of (dy * x), written by the consumer's weight-gradient pass ...
    bool ds_ready = false;      // ... when it ran before this attention's backward (then no pass over dy, x is needed)
    int att_in = -1, att_out = -1;
};

static bool defer_ok(Graph& g, int tid);        // (defined after ConvOp: the tensor's one convolution consumer is a plain Conv2D)

// ============================================================================================ Conv2D
struct ConvOp : GOp {
    int in, w, b, add, out, KS, Cout, relu, d2s;
    size_t wt_off = 0;
    bool add_grad_shared = false;   // the residual operand's gradient IS this op's (masked) output gradient: no copy
    AttFusion* att_after = nullptr;    // a ChannelAttention2D consumes this op's output (this op = its producer)
    AttFusion* att_before = nullptr;   // this op convolves the output of a ChannelAttention2D (this op = its consumer)
    ConvOp() { kind = "conv2d"; }
    // the convolved input: the tensor itself, or -- behind a fused attention -- the attention's INPUT seen through its scale
    TView in_view(Graph& g, int B, int bo, int bc) {
        if (att_before && att_before->fuse_scale) {
            TView v = g.view(att_before->att_in, B, false, bo, bc);
            const GTensor& t = g.tensors[in];
            v.sc = att_before->scale + (size_t)bo * t.nmul * t.C;
            return v;
        }
        return g.view(in, B, false, bo, bc);
    }
    // the output gradient: the tensor's gradient buffer, or -- under a fused attention backward -- the attention
    // output's gradient seen through dX = dY * scale + dmean
    TView dy_view(Graph& g, int B, int bo, int bc) {
        if (att_after && att_after->fuse_dx) {
            DL4DS_REQUIRE(d2s <= 1 && bo == 0 && (bc < 0 || bc == B), "fused attention backward: whole batches, no depth_to_space");
            TView v = g.view(att_after->att_out, B, true, bo, bc);
            v.sc = att_after->scale;
            v.sh = att_after->dmean;
            return v;
        }
        return out_view(g, true, B, bo, bc);
    }
    void on_prepare(Graph& g) override {
        if (add_grad_shared) g.tensors[add].grad = g.tensors[out].grad;
        plan_two_adds(g);
        if (add_grad_inplace) g.tensors[add].grad = g.tensors[out].grad;
    }
    void on_resolve_aliases(Graph& g) override {
        if (add >= 0 && (add_grad_shared || add_grad_inplace)) g.tensors[add].grad = g.tensors[out].grad;
    }
    // r feeds ONE convolution C and TWO fused adds A1 < A2 (a residual branch's input that is also its long skip: the
    // discriminator's branches, cfg5).  Until round 5: dZ(A2) was COPIED into r's gradient, dZ(A1) accumulated onto it, C's dgrad
    // accumulated again -- five passes over an 8 x 512^2 x 32 tensor beside the dgrad.  Now r's gradient buffer IS dZ(A1)'s (this op =
    // A1 decides, like add_grad_shared), dZ(A2) stays where it is as the pending operand of C's dgrad store (ConvOp::backward: dx =
    // dgrad + dZ(A2) + old value): no pass at all.  Needs C before A1 in op order (its backward comes last), equal shapes, no ReLU mask
    // on r's gradient, and no second stream (A1's weight gradient reads the buffer C's dgrad overwrites).
    bool add_grad_inplace = false;
    void plan_two_adds(Graph& g) {
        add_grad_inplace = false;
        if (add < 0 || add_grad_shared || d2s > 1 || g.aux_stream || att_before || att_after || test_env("DL4DS_NO_TWO_ADD_INPLACE")) return;
        GTensor& r = g.tensors[add];
        for (int o : g.outputs) if (o == add) return;
        if (r.is_input || !r.requires_grad || r.grad_masked || r.galias || r.alias_of >= 0 || r.n_conv_in != 1 || r.n_add_in != 0 ||
            r.n_masking != 0 || r.n_other != 2 || r.n_fused_add != 2 || r.per_sample() != g.tensors[out].per_sample())
            return;
        int me = -1, other = -1, consumer = -1;
        for (int i = 0; i < (int)g.ops.size(); ++i) {
            ConvOp* c = dynamic_cast<ConvOp*>(g.ops[i].get());
            if (!c) continue;
            if (c == this) me = i;
            else if (c->add == add) other = i;
            if (c->in == add) { if (c->att_before || c->att_after) return; consumer = i; }
        }
        if (me < 0 || other < 0 || consumer < 0) return;
        ConvOp* oc = static_cast<ConvOp*>(g.ops[other].get());
        if (oc->d2s > 1 || oc->add_grad_shared || oc->att_before || oc->att_after || g.tensors[oc->out].per_sample() != r.per_sample()) return;
        if (g.tensors[out].galias || g.tensors[out].alias_of >= 0) return;
        if (!(consumer < me && me < other)) return;        // this op is A1; A2 = `other` defers through r.two_add_inplace
        add_grad_inplace = true;
        r.two_add_inplace = true;
    }
    void on_finalize(Graph& g) override {
        // y = act(conv(x) + r): dL/dr = dZ.  When this add is r's only consumer (the 1x1-projected skip of a residual
        // block) r's gradient buffer can simply be dZ's -- nothing writes it again before r's producer has read it
        if (add >= 0 && d2s <= 1 && !exp_env("DL4DS_NO_GRAD_SHARE")) {
            const GTensor& r = g.tensors[add];
            bool is_output = false;
            for (int o : g.outputs) is_output |= (o == add);
            add_grad_shared = !r.is_input && !is_output && r.requires_grad && r.n_conv_in == 0 && r.n_add_in == 0 &&
                              r.n_masking == 0 && r.n_other == 1 && r.per_sample() == g.tensors[out].per_sample();
        }
        // ... and where it cannot be shared, the copy of dZ into r's gradient can apply r's own ReLU mask (r = ReLU output of a
        // Conv2D: the input of a residual block that also feeds the block's first convolution), which lets r's producer drop
        // its separate ReLU-backward pass as it does when all consumers are convolutions
        if (add >= 0 && !add_grad_shared && !exp_env("DL4DS_NO_MASK_FUSION")) {
            GTensor& r = g.tensors[add];
            bool is_output = false;
            for (int o : g.outputs) is_output |= (o == add);
            if (r.relu_out && !r.is_input && !is_output && r.n_other == r.n_fused_add) r.grad_masked = true;
        }
        wt_off = g.reserve_wt(g.params[w].n);
        g.add_wt_job(g.params[w].offset, false, wt_off, KS * KS, g.tensors[in].C, Cout);
        // ReLU backward fused into the consumers' dgrad stores when every consumer is a Conv2D reading this tensor
        // as its convolved input (DL4DS_NO_MASK_FUSION=1 keeps the separate pass, for A/B measurements)
        GTensor& t = g.tensors[out];
        bool is_output = false;
        for (int o : g.outputs) is_output |= (o == out);
        t.grad_masked = relu && !is_output && (t.n_conv_in + t.n_add_in + t.n_masking) >= 1 && t.n_other == 0 &&
                        !exp_env("DL4DS_NO_MASK_FUSION");
    }
    TView out_view(Graph& g, bool grad, int B, int bo, int bc) {
        const GTensor& ti = g.tensors[in];
        const GTensor& to = g.tensors[out];
        float* base = (grad ? to.grad : to.data) + (size_t)bo * to.per_sample();
        const int N = (bc < 0 ? B : bc) * to.nmul;
        if (d2s > 1) return make_view_d2s(base, N, ti.H, ti.W, Cout, d2s);
        if ((!grad || to.galias) && to.alias_of >= 0) return g.view(out, B, grad, bo, bc);       // straight into / out of a Concatenate's buffer
        return make_view(base, N, ti.H, ti.W, Cout);
    }
    void forward(Graph& g, int B, bool) override {
        ConvEpilogue ep;
        ep.bias = (b >= 0) ? g.wp(b) : nullptr;
        if (add >= 0) ep.add = g.view(add, B, false);
        ep.relu = relu;
        if (att_after && att_after->fuse_pool) ep.pool = att_after->pool;
        conv2d_forward(g.stream, in_view(g, B, 0, -1), g.wp(w), KS, out_view(g, false, B, 0, -1), ep);
    }
    size_t workspace_bytes(Graph& g, int B) override {
        TView x = g.view(in, B, false);
        TView dz = make_view(nullptr, x.N, x.H, x.W, Cout);
        const size_t att_ws = (size_t)(1024 + x.N) * ((size_t)KS * KS * x.C * Cout + Cout) * sizeof(float);   // per-image slabs
        return std::max(std::max(conv2d_wgrad_workspace_bytes(x, dz, KS), bias_grad_workspace_bytes(dz)), att_before ? att_ws : 0);
    }
    void backward(Graph& g, const BwdCtx& c) override {
        if (!g.tensors[out].grad_written) return;      // nothing flowed into this op
        TView dY = dy_view(g, c.B, c.b_off, c.b_cnt);
        if (relu && !g.tensors[out].grad_masked) {      // dZ = dY * [y > 0], in place (every consumer of y has contributed)
            DL4DS_REQUIRE(!dY.sc, "fused attention backward needs a linear producer");
            bias_act_backward(g.stream, dY, out_view(g, false, c.B, c.b_off, c.b_cnt), dY, nullptr, 0, g.workspace,
                              g.workspace_bytes);
        }
        if (add >= 0 && wants_grad(g, add, c)) {
            GTensor& ra = g.tensors[add];
            // r feeds exactly this add and ONE convolution (a residual block's input) and nothing has written its gradient yet:
            // leave dZ where it is; that convolution's dgrad store adds it (ConvOp::backward below), no copy
            bool defer = !add_grad_shared && !ra.grad_written && ra.n_conv_in == 1 && ra.n_add_in == 0 && ra.n_masking == 0 &&
                         ra.n_other == 1 && ra.n_fused_add == 1 && d2s <= 1 && !dY.sc && defer_ok(g, add) &&
                         !exp_env("DL4DS_NO_DEFERRED_ADD");
            // two fused adds planned copy-free (plan_two_adds): the LATER one (its backward runs first) leaves dZ as the pending
            // operand of the consuming convolution's dgrad store; the earlier one's dZ already is r's gradient buffer
            const bool inplace_here = add_grad_inplace && ra.two_add_inplace && !dY.sc;
            if (ra.two_add_inplace && !add_grad_inplace && !ra.grad_written && !ra.pending_add && d2s <= 1 && !dY.sc && defer_ok(g, add))
                defer = true;
            if (exp_env("DL4DS_ADD_DEBUG"))
                fprintf(stderr, "conv add grad: op out=%d add=%d defer=%d shared=%d written=%d n_conv_in=%d n_add_in=%d n_masking=%d n_other=%d n_fused_add=%d d2s=%d masked=%d C=%d H=%d B=%d\n",
                        out, add, (int)defer, (int)add_grad_shared, (int)ra.grad_written, ra.n_conv_in, ra.n_add_in, ra.n_masking, ra.n_other,
                        ra.n_fused_add, d2s, (int)ra.grad_masked, ra.C, ra.H, c.B);
            if (defer) {
                ra.pending_add = dY.p;
                ra.pending_view = dY;
            } else if (inplace_here && !ra.grad_written && dY.p == g.view(add, c.B, true, c.b_off, c.b_cnt).p) {
                // (r's gradient buffer IS this dZ: nothing to do)
            } else if (!add_grad_shared || inplace_here) {
                // (also the fall-back of a two-add plan whose buffers did not end up shared: the copying path is always right)
                if (g.tensors[add].grad_masked)
                    view_axpy_masked(g.stream, dY, g.view(add, c.B, false, c.b_off, c.b_cnt), g.view(add, c.B, true, c.b_off, c.b_cnt),
                                     g.tensors[add].grad_written);
                else
                    view_axpy(g.stream, dY, g.view(add, c.B, true, c.b_off, c.b_cnt), 1.f, g.tensors[add].grad_written);
            }
            if (!defer) g.tensors[add].grad_written = true;
        }
        if (c.param_grads) {     // weight gradient; the bias gradient (column sums of dZ) rides along
            const bool need_db = b >= 0;
            // on the aux stream: overlaps with this op's dgrad and everything after it on the main stream (dZ and
            // the input activation are not written again during this backward pass)
            g.fork_aux();
            const bool on_aux = aux_enabled(g);
            hipStream_t ws_stream = on_aux ? g.aux_stream : g.stream;
            float* ws_buf = on_aux ? g.aux_workspace : g.workspace;
            // behind a fused attention scale the stencil weight gradient is taken per image on the raw input: that yields
            // dW, db AND the attention's d(loss)/d(scale) (conv2d_direct_wgrad_attention), on the main stream
            bool done = false;
            if (att_before && att_before->fuse_scale && c.b_off == 0 && (c.b_cnt < 0 || c.b_cnt == c.B) &&
                !exp_env("DL4DS_NO_DS_FUSION")) {
                done = conv2d_direct_wgrad_attention(g.stream, g.view(att_before->att_in, c.B, false), dY, KS, att_before->scale,
                                                     g.wp(w), g.gp(w), g.params[w].grad_written, need_db ? g.gp(b) : nullptr,
                                                     need_db ? (int)g.params[b].grad_written : 0, att_before->ds, g.workspace,
                                                     g.workspace_bytes);
                att_before->ds_ready = done;
            }
            if (!done)
                conv2d_wgrad(ws_stream, in_view(g, c.B, c.b_off, c.b_cnt), dY, KS, g.gp(w),
                             g.params[w].grad_written, need_db ? g.gp(b) : nullptr,
                             need_db ? (int)g.params[b].grad_written : 0, ws_buf, g.workspace_bytes);
            g.params[w].grad_written = true;
            if (need_db) g.params[b].grad_written = true;
        }
        if (wants_grad(g, in, c)) {
            float* wt = g.Wt + wt_off;              // (filled by Graph::refresh_dgrad_weights at the start of this pass)
            ConvEpilogue ep;
            ep.accumulate = g.tensors[in].grad_written;
            // the producer's ReLU backward rides on this store (saves a read-modify-write pass over the gradient)
            if (g.tensors[in].grad_masked) ep.mask = g.view(in, c.B, false, c.b_off, c.b_cnt);
            if (g.tensors[in].pending_add) {
                // the gradient that reached this tensor through the residual add of its block: dx = dgrad + dZ_add (then masked)
                ep.add = g.tensors[in].pending_view;       // (its own pixel pitch: dZ may live inside a Concatenate's gradient buffer)
                g.tensors[in].pending_add = nullptr;
            }
            conv2d_forward(g.stream, dY, wt, KS, g.view(in, c.B, true, c.b_off, c.b_cnt), ep);
            g.tensors[in].grad_written = true;
        }
    }
};

static bool defer_ok(Graph& g, int tid) {
    // cached per tensor would be nicer; graphs have a few hundred ops and this runs once per residual block and step
    for (auto& up : g.ops)
        if (ConvOp* c = dynamic_cast<ConvOp*>(up.get()))
            if (c->in == tid) return !c->att_before && !c->att_after;
    return false;                                     // (the consumer is a folded / transposed / other convolution)
}

// ============================================================================================ ChannelAttention
struct ChAttOp : GOp {
    int in, out, w1, b1, w2, b2, Cr, T5;   // T5 > 0: 5-D mode (B,T,H,W,C), mean over (T,H)
    AttFusion fz;
    ConvOp* producer = nullptr;            // ConvOp writing `in` (found at finalize), ConvOp reading `out`
    ConvOp* consumer = nullptr;
    ChAttOp() { kind = "chatt"; }
    AttShape shape(Graph& g, int B) {
        const GTensor& t = g.tensors[in];
        AttShape s;
        if (T5 > 0) { s.G = B; s.R = t.nmul * t.H; s.P = t.W; }
        else { s.G = B * t.nmul; s.R = t.H * t.W; s.P = 1; }
        s.C = t.C; s.Cr = Cr;
        return s;
    }
    int pool_tiles(Graph& g) { const GTensor& t = g.tensors[in]; return conv2d_narrow_pair_tiles_per_image(t.H, t.W); }
    size_t saved_floats_per_sample(Graph& g) override {
        const GTensor& t = g.tensors[in];
        const size_t inst = (T5 > 0) ? (size_t)t.W : (size_t)t.nmul;
        // mean, scale, dmean, ds (C each) + hidden (Cr) per instance; 4-D mode: room for the producer's pooling partial sums
        return inst * (4 * (size_t)t.C + Cr) + (T5 > 0 ? 0 : (size_t)t.nmul * pool_tiles(g) * 8 + 16);
    }
    size_t workspace_bytes(Graph& g, int B) override { return chatt_workspace_bytes(shape(g, B)); }
    void ptrs(Graph& g, int, float*& mean, float*& hidden, float*& scale, float*& dmean, float*& pool, float** ds_out = nullptr) {
        // laid out for the LARGEST batch the buffers were allocated for, so that the addresses handed to the neighbouring
        // convolutions (AttFusion) stay put when a smaller batch runs
        AttShape s = shape(g, g.maxB);
        const size_t inst = (size_t)s.G * s.P;
        mean = saved; scale = mean + inst * s.C; hidden = scale + inst * s.C; dmean = hidden + inst * s.Cr;
        float* dsp = dmean + inst * s.C;
        if (ds_out) *ds_out = dsp;
        pool = dsp + ((inst * s.C + 3) & ~(size_t)3);          // keep the float4 records 16-byte aligned
    }
    void on_finalize(Graph& g) override {
        // neighbours: the convolution producing `in` and the (single) convolution reading `out`
        for (auto& op : g.ops) {
            ConvOp* c = dynamic_cast<ConvOp*>(op.get());
            if (!c) continue;
            if (c->out == in) producer = c;
            if (c->in == out) consumer = c;
        }
        fz.att_in = in; fz.att_out = out;
        if (producer) producer->att_after = &fz;           // (the fuse_* flags stay false until on_prepare has seen the buffers)
        if (consumer) consumer->att_before = &fz;
    }
    // which pieces can be handed to the neighbours: needs the real views (alignment), so decided once the buffers exist
    void on_prepare(Graph& g) override {
        fz.fuse_pool = fz.fuse_scale = fz.fuse_dx = false;
        if (T5 > 0 || test_env("DL4DS_NO_TAIL_FUSION")) return;
        const int B = g.maxB;
        float *mean, *hidden, *scale, *dmean, *pool;
        ptrs(g, B, mean, hidden, scale, dmean, pool, &fz.ds);
        fz.scale = scale; fz.dmean = dmean; fz.pool = pool; fz.pool_tiles = pool_tiles(g);
        const GTensor& ti = g.tensors[in];
        const GTensor& to = g.tensors[out];
        auto is_output = [&](int t) { for (int o : g.outputs) if (o == t) return true; return false; };
        const bool in_private = !is_output(in) && ti.n_conv_in == 0 && ti.n_add_in == 0 && ti.n_masking == 0 && ti.n_other == 1;     // only this op reads it
        // ---- pooling partials from the producer's epilogue
        if (producer && producer->d2s <= 1 && ti.C <= 8) {
            ConvEpilogue ep;
            ep.bias = producer->b >= 0 ? g.wp(producer->b) : nullptr;
            if (producer->add >= 0) ep.add = g.view(producer->add, B, false);
            TView pin = producer->in_view(g, B, 0, -1), pout = g.view(in, B, false);
            fz.fuse_pool = !conv2d_direct_eligible(pin, pout, producer->KS) && conv2d_narrow_pair_ok(pin, pout, producer->KS, ep);
        }
        // ---- scale applied by the consumer's loads: `out` has exactly one reader, a 3x3 convolution on the stencil kernels
        if (consumer && !is_output(out) && to.n_conv_in == 1 && to.n_add_in == 0 && to.n_masking == 0 && to.n_other == 0 && consumer->KS == 3) {
            TView x = g.view(in, B, false);
            x.sc = scale;
            TView y = make_view(nullptr, x.N, x.H, x.W, consumer->Cout);
            fz.fuse_scale = conv2d_direct_eligible(x, y, 3);
        }
        // ---- dX applied by the producer's backward loads: linear producer without a residual operand, narrow kernels
        if (producer && in_private && producer->relu == 0 && producer->add < 0 && producer->d2s <= 1 && producer->KS == 3 &&
            !ti.grad_masked) {
            TView dz = g.view(out, B, true);
            dz.sc = scale; dz.sh = dmean;
            TView px = producer->in_view(g, B, 0, -1);
            TView dz_plain = dz;
            dz_plain.sc = dz_plain.sh = nullptr;
            // weight gradient: the narrow kernel takes the affine on its dz operand (the stencil wgrad does not)
            bool ok = !px.sc && !conv2d_direct_eligible(px, dz_plain, 3) && conv2d_narrow_wgrad_slabs(px, dz, 3) > 0;
            // input gradient = convolution of dY: stencil or pair kernel, both read the affine
            ConvEpilogue ep;
            const GTensor& tpi = g.tensors[producer->in];
            TView dx = make_view(tpi.grad ? tpi.grad : tpi.data, px.N, px.H, px.W, px.C);
            if (tpi.grad_masked) ep.mask = g.view(producer->in, B, false);
            ok = ok && (conv2d_direct_eligible(dz, dx, 3) ||
                        (!conv2d_direct_eligible(dz_plain, dx, 3) && conv2d_narrow_pair_ok(dz, dx, 3, ep)));
            fz.fuse_dx = ok;
        }
    }
    void forward(Graph& g, int B, bool) override {
        float *mean, *hidden, *scale, *dmean, *pool;
        ptrs(g, B, mean, hidden, scale, dmean, pool);
        chatt_forward(g.stream, g.tensors[in].data, fz.fuse_scale ? nullptr : g.tensors[out].data, shape(g, B), g.wp(w1),
                      g.wp(b1), g.wp(w2), g.wp(b2), mean, hidden, scale, g.workspace, fz.fuse_pool ? pool : nullptr,
                      fz.pool_tiles);
    }
    std::string describe_fusion(Graph&) override {
        return std::string("{\"op\":\"chatt\",\"in\":") + std::to_string(in) + ",\"pool_from_producer\":" +
               (fz.fuse_pool ? "true" : "false") + ",\"scale_in_consumer_load\":" + (fz.fuse_scale ? "true" : "false") +
               ",\"dx_in_producer_backward\":" + (fz.fuse_dx ? "true" : "false") + ",\"dscale_from_consumer_wgrad\":" +
               (fz.fuse_scale && !exp_env("DL4DS_NO_DS_FUSION") ? "true" : "false") + "}";
    }
    bool partial_batch_ok() const override { return false; }
    void backward(Graph& g, const BwdCtx& c) override {
        if (!g.tensors[out].grad_written) return;
        DL4DS_REQUIRE(c.b_off == 0 && (c.b_cnt < 0 || c.b_cnt == c.B), "chatt: partial-batch backward not supported");
        float *mean, *hidden, *scale, *dmean, *pool;
        ptrs(g, c.B, mean, hidden, scale, dmean, pool);
        const bool dx = wants_grad(g, in, c);
        DL4DS_REQUIRE(dx, "chatt: input must require grad");
        const int accw = g.params[w1].grad_written;
        chatt_backward(g.stream, g.tensors[in].data, g.tensors[out].grad, fz.fuse_dx ? nullptr : g.tensors[in].grad,
                       g.tensors[in].grad_written, shape(g, c.B), g.wp(w1), g.wp(w2), mean, hidden, scale,
                       c.param_grads ? g.gp(w1) : nullptr, g.gp(b1), g.gp(w2), g.gp(b2), accw, g.workspace, dmean,
                       fz.ds_ready ? fz.ds : nullptr);
        fz.ds_ready = false;
        g.tensors[in].grad_written = true;
        if (c.param_grads) {
            g.params[w1].grad_written = g.params[b1].grad_written = true;
            g.params[w2].grad_written = g.params[b2].grad_written = true;
        }
    }
};

// ============================================================================================ Concat
struct ConcatOp : GOp {
    std::vector<int> ins;
    int out;
    ConcatOp() { kind = "concat"; }
    TView slice(Graph& g, int B, bool grad, int k, int bo, int bc) {
        TView v = g.view(out, B, grad, bo, bc);
        int off = 0;
        for (int i = 0; i < k; ++i) off += g.tensors[ins[i]].C;
        v.p += off;
        v.C = g.tensors[ins[k]].C;
        v.vec = v.vec && ((off & 3) == 0) && ((v.C & 3) == 0);
        return v;
    }
    void forward(Graph& g, int B, bool) override {
        // two or more dense inputs that need a copy into a dense output: one pass that writes the wide tensor contiguously
        {
            const TView wide = g.view(out, B, false);
            ConcatSlice sl[4];
            int n = 0, off = 0;
            bool ok = wide.d2s <= 1 && wide.ld == wide.C && wide.nstride == (size_t)wide.H * wide.W * wide.C && !exp_env("DL4DS_NO_CONCAT_JOIN");
            for (size_t k = 0; k < ins.size() && ok; ++k) {
                const GTensor& ti = g.tensors[ins[k]];
                if (ti.alias_parent != out) {
                    const TView d = g.view(ins[k], B, false);
                    ok = n < 4 && d.d2s <= 1 && d.ld == d.C && d.nstride == (size_t)d.H * d.W * d.C && !d.sc;
                    if (ok) sl[n++] = ConcatSlice{d.p, nullptr, off, ti.C, 0};
                }
                off += ti.C;
            }
            if (ok && n >= 2) {
                concat_join(g.stream, wide.p, wide.ld, (size_t)wide.N * wide.H * wide.W, sl, n);
                return;
            }
        }
        for (size_t k = 0; k < ins.size(); ++k) {
            if (g.tensors[ins[k]].alias_parent == out) continue;     // its producer wrote it here already
            view_axpy(g.stream, g.view(ins[k], B, false), slice(g, B, false, (int)k, 0, -1), 1.f, 0);
        }
    }
    void backward(Graph& g, const BwdCtx& c) override {
        if (!g.tensors[out].grad_written) return;
        // two or more inputs that need a copy, everything dense: one pass over the wide gradient (concat_split)
        {
            const TView wide = g.view(out, c.B, true, c.b_off, c.b_cnt);
            const bool wide_plain = wide.d2s <= 1 && wide.ld == wide.C && wide.nstride == (size_t)wide.H * wide.W * wide.C;
            ConcatSlice sl[4];
            int n = 0, off = 0;
            bool ok = wide_plain && !exp_env("DL4DS_NO_CONCAT_SPLIT");
            for (size_t k = 0; k < ins.size() && ok; ++k) {
                const GTensor& ti = g.tensors[ins[k]];
                if (wants_grad(g, ins[k], c) && !ti.galias) {
                    const TView d = g.view(ins[k], c.B, true, c.b_off, c.b_cnt);
                    ok = n < 4 && d.d2s <= 1 && d.ld == d.C && d.nstride == (size_t)d.H * d.W * d.C;
                    const float* mk = nullptr;
                    if (ok && ti.grad_masked) {
                        const TView m = g.view(ins[k], c.B, false, c.b_off, c.b_cnt);
                        ok = m.d2s <= 1 && m.ld == m.C && m.nstride == (size_t)m.H * m.W * m.C && !m.sc;
                        mk = m.p;
                    }
                    if (ok) sl[n++] = ConcatSlice{d.p, mk, off, ti.C, ti.grad_written ? 1 : 0};
                }
                off += ti.C;
            }
            if (ok && n >= 2) {
                concat_split(g.stream, wide.p, wide.ld, (size_t)wide.N * wide.H * wide.W, sl, n);
                for (size_t k = 0; k < ins.size(); ++k)
                    if (wants_grad(g, ins[k], c)) g.tensors[ins[k]].grad_written = true;
                return;
            }
        }
        for (size_t k = 0; k < ins.size(); ++k) {
            if (!wants_grad(g, ins[k], c)) continue;
            if (g.tensors[ins[k]].galias) { g.tensors[ins[k]].grad_written = true; continue; }     // its gradient IS this slice
            // a ReLU output whose consumers apply the mask: this copy zeroes the gradient where the activation is <= 0
            TView mask{nullptr, 0, 0, 0, 0, 0, 0, 0};
            if (g.tensors[ins[k]].grad_masked) mask = g.view(ins[k], c.B, false, c.b_off, c.b_cnt);
            view_axpy_masked(g.stream, slice(g, c.B, true, (int)k, c.b_off, c.b_cnt), mask,
                             g.view(ins[k], c.B, true, c.b_off, c.b_cnt), g.tensors[ins[k]].grad_written);
            g.tensors[ins[k]].grad_written = true;
        }
    }
};

// ============================================================================================ Add (+ReLU)
struct AddOp : GOp {
    int a, b, out, relu;
    AddOp() { kind = "add"; }
    void forward(Graph& g, int B, bool) override {
        add_act(g.stream, g.tensors[a].data, g.tensors[b].data, g.tensors[out].data,
                g.tensors[out].per_sample() * B, relu);
    }
    void backward(Graph& g, const BwdCtx& c) override {
        if (!g.tensors[out].grad_written) return;
        TView dY = g.view(out, c.B, true, c.b_off, c.b_cnt);
        if (relu) {
            TView none{nullptr, 0, 0, 0, 0, 0, 0, 0};
            bias_act_backward(g.stream, dY, g.view(out, c.B, false, c.b_off, c.b_cnt), dY, nullptr, 0, g.workspace,
                              g.workspace_bytes);
        }
        if (a != b && wants_grad(g, a, c) && wants_grad(g, b, c) && g.tensors[a].grad_masked && g.tensors[b].grad_masked &&
            !exp_env("DL4DS_NO_MASKED_PAIR")) {
            // both operands are ReLU outputs whose masks ride on this copy: one pass that reads dY once (cfg2's long skip)
            const size_t ps = g.tensors[a].per_sample();
            const size_t off = (size_t)c.b_off * ps, n = (size_t)(c.b_cnt < 0 ? c.B : c.b_cnt) * ps;
            if (g.tensors[b].per_sample() == ps &&
                masked_axpy_pair(g.stream, g.tensors[out].grad + off, g.tensors[a].data + off, g.tensors[a].grad + off, g.tensors[a].grad_written,
                                 g.tensors[b].data + off, g.tensors[b].grad + off, g.tensors[b].grad_written, n)) {
                g.tensors[a].grad_written = true;
                g.tensors[b].grad_written = true;
                return;
            }
        }
        for (int t : {a, b}) {
            if (!wants_grad(g, t, c)) continue;
            if (exp_env("DL4DS_ADD_DEBUG"))
                fprintf(stderr, "add op grad: out=%d operand=%d masked=%d written=%d C=%d H=%d B=%d\n", out, t, (int)g.tensors[t].grad_masked,
                        (int)g.tensors[t].grad_written, g.tensors[t].C, g.tensors[t].H, c.B);
            if (g.tensors[t].grad_masked) {
                // the operand is a ReLU output whose mask its consumers apply: fold it into this copy
                const size_t ps = g.tensors[t].per_sample();
                const size_t off = (size_t)c.b_off * ps, n = (size_t)(c.b_cnt < 0 ? c.B : c.b_cnt) * ps;
                masked_axpy(g.stream, g.tensors[out].grad + off, g.tensors[t].data + off, g.tensors[t].grad + off, n,
                            g.tensors[t].grad_written);
            } else {
                view_axpy(g.stream, dY, g.view(t, c.B, true, c.b_off, c.b_cnt), 1.f, g.tensors[t].grad_written);
            }
            g.tensors[t].grad_written = true;
        }
    }
};

// ============================================================================================ Activation
struct ActOp : GOp {
    int in, out, act;
    ActOp() { kind = "act"; }
    void forward(Graph& g, int B, bool) override {
        act_forward(g.stream, g.tensors[in].data, g.tensors[out].data, g.tensors[out].per_sample() * B, act);
    }
    void backward(Graph& g, const BwdCtx& c) override {
        if (!g.tensors[out].grad_written || !wants_grad(g, in, c)) return;
        const size_t ps = g.tensors[in].per_sample();
        const size_t off = (size_t)c.b_off * ps;
        const size_t n = (size_t)(c.b_cnt < 0 ? c.B : c.b_cnt) * ps;
        act_backward(g.stream, g.tensors[in].data + off, g.tensors[out].grad + off, g.tensors[in].grad + off, n, act,
                     g.tensors[in].grad_written);
        g.tensors[in].grad_written = true;
    }
};

// ============================================================================================ MaxPool 2x2
struct MaxPoolOp : GOp {
    int in, out;
    MaxPoolOp() { kind = "maxpool2"; }
    void forward(Graph& g, int B, bool) override {
        maxpool2_forward(g.stream, g.view(in, B, false), g.view(out, B, false));
    }
    void backward(Graph& g, const BwdCtx& c) override {
        if (!g.tensors[out].grad_written || !wants_grad(g, in, c)) return;
        const GTensor& t = g.tensors[in];
        TView dx = g.view(in, c.B, true, c.b_off, c.b_cnt);
        int acc = t.grad_written;
        if (!acc && ((t.H | t.W) & 1)) {    // rows/cols dropped by VALID pooling receive zero gradient
            fill(g.stream, dx.p, (size_t)dx.N * dx.H * dx.W * dx.C, 0.f);
            acc = 1;
        }
        maxpool2_backward(g.stream, g.view(in, c.B, false, c.b_off, c.b_cnt), g.view(out, c.B, false, c.b_off, c.b_cnt),
                          g.view(out, c.B, true, c.b_off, c.b_cnt), dx, acc, t.grad_masked ? 1 : 0);
        g.tensors[in].grad_written = true;
    }
};

// ============================================================================================ Bilinear resize
// Index / weight tables of tf.image.resize(method='bicubic') along one axis (ResizeBicubic, half_pixel_centers=True), in the
// op's own float arithmetic: scale = in / out; src = (o + 0.5f) * scale - 0.5f; i0 = floor(src); the Keys (A = -0.5) weights
// come from a 1024-step table at offset = lrintf((src - i0) * 1024); a tap whose index had to be clamped into the image gets
// weight 0 and the remaining weights are renormalised to sum 1.
// ... and of tf.image.resize(method='bilinear') (half-pixel centres), the arithmetic of elementwise.hip's bilinear_src: two taps
// lo = max(floor(src), 0), hi = min(ceil(src), in - 1) with weights 1 - f, f (f = src - floor(src)).  The table kernels evaluate
// them with four channels per thread and, in the backward pass, as the exact transpose (CSR gather) -- the direct kernels spent
// ~3 600 instructions per input element on re-deriving the taps of the ~120 candidate outputs of a x4 up-sampling.
static void bilinear_axis_tables(int in_size, int out_size, std::vector<int>& idx, std::vector<float>& w) {
    const float scale = (float)in_size / (float)out_size;
    idx.assign((size_t)out_size * 2, 0);
    w.assign((size_t)out_size * 2, 0.f);
    for (int o = 0; o < out_size; ++o) {
        const float src = ((float)o + 0.5f) * scale - 0.5f;
        const float fl = std::floor(src);
        const float f = src - fl;
        idx[(size_t)o * 2] = std::max((int)fl, 0);
        idx[(size_t)o * 2 + 1] = std::min((int)std::ceil(src), in_size - 1);
        w[(size_t)o * 2] = 1.f - f;
        w[(size_t)o * 2 + 1] = f;
    }
}

static void bicubic_axis_tables(int in_size, int out_size, std::vector<int>& idx, std::vector<float>& w) {
    const int T = 1024;
    std::vector<float> lut((T + 1) * 2);
    const float A = -0.5f;
    for (int i = 0; i <= T; ++i) {
        float x = i * 1.0f / T;
        lut[i * 2] = ((A + 2) * x - (A + 3)) * x * x + 1;
        x += 1.0f;
        lut[i * 2 + 1] = ((A * x - 5 * A) * x + 8 * A) * x - 4 * A;
    }
    const float scale = (float)in_size / (float)out_size;
    idx.assign((size_t)out_size * 4, 0);
    w.assign((size_t)out_size * 4, 0.f);
    for (int o = 0; o < out_size; ++o) {
        const float src = ((float)o + 0.5f) * scale - 0.5f;
        const long i0 = (long)std::floor(src);
        const float delta = src - (float)i0;
        const long off = lrintf(delta * T);
        const float wt[4] = {lut[off * 2 + 1], lut[off * 2], lut[(T - off) * 2], lut[(T - off) * 2 + 1]};
        float sum = 0.f;
        for (int k = 0; k < 4; ++k) {
            const long want = i0 - 1 + k;
            const long got = std::min<long>(std::max<long>(want, 0), in_size - 1);
            idx[(size_t)o * 4 + k] = (int)got;
            w[(size_t)o * 4 + k] = (got == want) ? wt[k] : 0.f;
            sum += w[(size_t)o * 4 + k];
        }
        if (std::fabs(sum) >= 1000.f * std::numeric_limits<float>::min())
            for (int k = 0; k < 4; ++k) w[(size_t)o * 4 + k] *= 1.0f / sum;
    }
}

// The same for tf.image.resize(method = 'lanczos3' | 'lanczos5' | 'gaussian' | 'mitchellcubic', antialias=False), i.e.
// ScaleAndTranslate with scale = out / in, translation 0 and kernel scale 1 (its ComputeSpans, in the op's float arithmetic):
// sample = (o + 0.5f) / scale; taps ceil(sample - R - 0.5) .. floor(sample + R - 0.5) clamped into the image, weight
// kernel(|i + 0.5 - sample|), normalised to sum 1.  Kernels (sampling_kernels.h): Lanczos radius R: 0 beyond R, 1 within 1e-3,
// else R sin(pi x) sin(pi x / R) / (pi x)^2; Gaussian: radius 1.5, sigma 0.5; Mitchell-Netravali cubic (B = C = 1/3), radius 2.
// Returns the number of taps per output (the widest span); narrower spans are padded with weight 0.
static int scale_translate_axis_tables(int method, int in_size, int out_size, std::vector<int>& idx, std::vector<float>& w) {
    const float R = method == 3 ? 3.f : method == 4 ? 5.f : method == 5 ? 1.5f : 2.f;
    auto kernel = [&](float x) -> float {
        x = std::fabs(x);
        if (method == 3 || method == 4) {
            const float kPI = 3.14159265359f;
            if (x > R) return 0.f;
            if (x <= 1e-3f) return 1.f;
            return R * std::sin(kPI * x) * std::sin(kPI * x / R) / (kPI * kPI * x * x);
        }
        if (method == 5) {
            const float sigma = R / 3.f;
            if (x >= R) return 0.f;
            return std::exp(-x * x / (2.0f * sigma * sigma));
        }
        if (x >= 2.f) return 0.f;
        if (x >= 1.f) return (((-7.0f / 18.0f) * x + 2.0f) * x - 10.0f / 3.0f) * x + 16.0f / 9.0f;
        return (((7.0f / 6.0f) * x - 2.0f) * x) * x + 8.0f / 9.0f;
    };
    const float inv_scale = 1.0f / ((float)out_size / (float)in_size);
    std::vector<int> start(out_size, 0);
    std::vector<std::vector<float>> ws(out_size);
    int K = 1;
    for (int o = 0; o < out_size; ++o) {
        const float sample = ((float)o + 0.5f) * inv_scale;
        if (sample < 0 || sample > (float)in_size) continue;
        long s0 = (long)std::ceil(sample - R - 0.5f), s1 = (long)std::floor(sample + R - 0.5f);
        s0 = std::min<long>(std::max<long>(s0, 0), in_size - 1);
        s1 = std::min<long>(std::max<long>(s1, 0), in_size - 1) + 1;
        float tot = 0.f;
        for (long i = s0; i < s1; ++i) {
            const float wt = kernel((float)i + 0.5f - sample);
            tot += wt;
            ws[o].push_back(wt);
        }
        if (std::fabs(tot) >= 1000.f * std::numeric_limits<float>::min())
            for (float& v : ws[o]) v *= 1.0f / tot;
        start[o] = (int)s0;
        K = std::max<int>(K, (int)ws[o].size());
    }
    idx.assign((size_t)out_size * K, 0);
    w.assign((size_t)out_size * K, 0.f);
    for (int o = 0; o < out_size; ++o)
        for (size_t k = 0; k < ws[o].size(); ++k) { idx[(size_t)o * K + k] = start[o] + (int)k; w[(size_t)o * K + k] = ws[o][k]; }
    return K;
}

struct ResizeOp : GOp {
    int in, out;
    bool nearest = false, bicubic = false;      // `bicubic`: any table-driven method
    int method = 0, ky = 4, kx = 4;             // 0 bilinear 1 nearest 2 bicubic 3 lanczos3 4 lanczos5 5 gaussian 6 mitchellcubic
    // table-driven: forward tables [out][k] and their transpose as CSR over the input index, on the device
    int *d_iy = nullptr, *d_ix = nullptr, *d_py = nullptr, *d_oy = nullptr, *d_px = nullptr, *d_ox = nullptr;
    int max_back_x = 0;          // most output columns any input column feeds (the backward gather's row length)
    float *d_wy = nullptr, *d_wx = nullptr, *d_vy = nullptr, *d_vx = nullptr;
    ResizeOp() { kind = "resize"; }
    ~ResizeOp() override {
        for (void* p : {(void*)d_iy, (void*)d_ix, (void*)d_py, (void*)d_oy, (void*)d_px, (void*)d_ox, (void*)d_wy, (void*)d_wx,
                        (void*)d_vy, (void*)d_vx})
            if (p) (void)hipFree(p);
    }
    template <class T>
    static T* upload(const std::vector<T>& v) {
        T* d = nullptr;
        HIP_CHECK(hipMalloc((void**)&d, std::max<size_t>(v.size(), 1) * sizeof(T)));
        if (!v.empty()) HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
        return d;
    }
    int axis(int in_size, int out_size, int*& d_i, float*& d_w, int*& d_p, int*& d_o, float*& d_v, int& max_back) const {
        std::vector<int> idx;
        std::vector<float> w;
        int K = 4;
        if (method == 2) bicubic_axis_tables(in_size, out_size, idx, w);
        else if (method == 0) { bilinear_axis_tables(in_size, out_size, idx, w); K = 2; }
        else K = scale_translate_axis_tables(method, in_size, out_size, idx, w);
        // transpose: per input index the (output, weight) pairs, output index ascending, tap ascending (fixed summation order)
        std::vector<std::vector<std::pair<int, float>>> cols(in_size);
        for (int o = 0; o < out_size; ++o)
            for (int k = 0; k < K; ++k)
                if (w[(size_t)o * K + k] != 0.f) cols[idx[(size_t)o * K + k]].push_back({o, w[(size_t)o * K + k]});
        std::vector<int> ptr(in_size + 1, 0), oo;
        std::vector<float> vv;
        for (int i = 0; i < in_size; ++i) {
            for (auto& pr : cols[i]) { oo.push_back(pr.first); vv.push_back(pr.second); }
            ptr[i + 1] = (int)oo.size();
            max_back = std::max(max_back, (int)cols[i].size());
        }
        d_i = upload(idx); d_w = upload(w); d_p = upload(ptr); d_o = upload(oo); d_v = upload(vv);
        return K;
    }
    void on_finalize(Graph& g) override {
        if (!bicubic) return;
        int my = 0;
        max_back_x = 0;
        ky = axis(g.tensors[in].H, g.tensors[out].H, d_iy, d_wy, d_py, d_oy, d_vy, my);
        kx = axis(g.tensors[in].W, g.tensors[out].W, d_ix, d_wx, d_px, d_ox, d_vx, max_back_x);
    }
    void forward(Graph& g, int B, bool) override {
        if (bicubic) resize_table_forward(g.stream, g.view(in, B, false), g.view(out, B, false), d_iy, d_wy, d_ix, d_wx, ky, kx);
        else if (nearest) resize_nearest_forward(g.stream, g.view(in, B, false), g.view(out, B, false));
        else resize_bilinear_forward(g.stream, g.view(in, B, false), g.view(out, B, false));
    }
    void backward(Graph& g, const BwdCtx& c) override {
        if (!g.tensors[out].grad_written || !wants_grad(g, in, c)) return;
        if (bicubic)
            resize_table_backward(g.stream, g.view(out, c.B, true, c.b_off, c.b_cnt), g.view(in, c.B, true, c.b_off, c.b_cnt), d_py, d_oy,
                                  d_vy, d_px, d_ox, d_vx, g.tensors[in].grad_written, max_back_x);
        else
            (nearest ? resize_nearest_backward : resize_bilinear_backward)(g.stream, g.view(out, c.B, true, c.b_off, c.b_cnt),
                                     g.view(in, c.B, true, c.b_off, c.b_cnt), g.tensors[in].grad_written);
        g.tensors[in].grad_written = true;
    }
};

// ============================================================================================ LocallyConnected 1x1
struct LocalConvOp : GOp {
    int in, out, w, b;
    LocalConvOp() { kind = "localconv"; }
    int alias_output() const override { return out; }      // stores through a view: may live inside the Concatenate that follows
    void forward(Graph& g, int B, bool) override {
        localconv_forward(g.stream, g.view(in, B, false), g.wp(w), b >= 0 ? g.wp(b) : nullptr, g.view(out, B, false));
    }
    void backward(Graph& g, const BwdCtx& c) override {
        if (!g.tensors[out].grad_written) return;
        DL4DS_REQUIRE(c.param_grads, "localconv: backward without parameter gradients not supported");
        const bool dx = wants_grad(g, in, c);
        TView dxv{nullptr, 0, 0, 0, 0, 0, 0, 0};
        if (dx) dxv = g.view(in, c.B, true, c.b_off, c.b_cnt);
        localconv_backward(g.stream, g.view(in, c.B, false, c.b_off, c.b_cnt), g.wp(w),
                           g.view(out, c.B, true, c.b_off, c.b_cnt), dxv, g.tensors[in].grad_written, g.gp(w),
                           b >= 0 ? g.gp(b) : nullptr, g.params[w].grad_written);
        if (dx) g.tensors[in].grad_written = true;
        g.params[w].grad_written = true;
        if (b >= 0) g.params[b].grad_written = true;
    }
};

// ============================================================================================ expand+repeat over T
// tf.expand_dims(s,1); tf.repeat(s, T, axis=1)  (spt_postups.py:139-140): (B,H,W,C) -> (B,T,H,W,C)
struct RepeatTimeOp : GOp {
    int in, out, T;
    RepeatTimeOp() { kind = "repeat_time"; }
    int alias_output() const override { return out; }
    void forward(Graph& g, int B, bool) override {
        if (g.tensors[out].alias_of >= 0) {                  // straight into the Concatenate's buffer
            DL4DS_REQUIRE(g.tensors[in].alias_of < 0, "repeat_time: dense input expected");
            repeat_time_forward_view(g.stream, g.tensors[in].data, g.view(out, B, false), B, T);
            return;
        }
        repeat_time_forward(g.stream, g.tensors[in].data, g.tensors[out].data, B, T, g.tensors[in].per_sample());
    }
    void backward(Graph& g, const BwdCtx& c) override {
        if (!g.tensors[out].grad_written || !wants_grad(g, in, c)) return;
        const GTensor& ti = g.tensors[in];
        const size_t ps = ti.per_sample();
        const int cnt = c.b_cnt < 0 ? c.B : c.b_cnt;
        if (g.tensors[out].galias)           // the gradient is a channel slice of the Concatenate's
            repeat_time_backward_view(g.stream, g.view(out, c.B, true, c.b_off, c.b_cnt), ti.grad + (size_t)c.b_off * ps, cnt, T, ti.grad_written);
        else
            repeat_time_backward(g.stream, g.tensors[out].grad + (size_t)c.b_off * T * ps, ti.grad + (size_t)c.b_off * ps, cnt, T, ps,
                                 ti.grad_written);
        g.tensors[in].grad_written = true;
    }
};

template <class T>
T* push(Graph& g) {
    T* p = new T();
    g.ops.emplace_back(p);
    return p;
}

}  // namespace

// ============================================================================================ constructors
int g_conv2d(Graph& g, int in, int w, int b, int add, int KS, int Cout, int relu, int d2s) {
    const GTensor ti = g.tensors.at(in);
    DL4DS_REQUIRE(KS == 1 || KS == 3 || KS == 5 || KS == 7, "conv2d: kernel size must be 1, 3, 5 or 7");
    DL4DS_REQUIRE(g.params.at(w).n == (size_t)KS * KS * ti.C * Cout, "conv2d: kernel parameter size mismatch");
    if (b >= 0) DL4DS_REQUIRE(g.params.at(b).n == (size_t)Cout, "conv2d: bias size mismatch");
    int out;
    if (d2s > 1) {
        DL4DS_REQUIRE(Cout % (d2s * d2s) == 0, "conv2d: Cout not divisible by r^2");
        DL4DS_REQUIRE(add < 0, "conv2d: residual add cannot be combined with depth_to_space");
        out = g.add_tensor(ti.H * d2s, ti.W * d2s, Cout / (d2s * d2s), ti.nmul, true, false);
    } else {
        out = g.add_tensor(ti.H, ti.W, Cout, ti.nmul, true, false);
    }
    if (add >= 0) {
        const GTensor& ta = g.tensors.at(add);
        DL4DS_REQUIRE(ta.H == ti.H && ta.W == ti.W && ta.C == Cout && ta.nmul == ti.nmul, "conv2d: add shape mismatch");
    }
    ConvOp* op = push<ConvOp>(g);
    op->in = in; op->w = w; op->b = b; op->add = add; op->out = out; op->KS = KS; op->Cout = Cout; op->relu = relu;
    op->pids = {w, b};
    g.tensors[in].n_conv_in++;
    if (add >= 0) { g.tensors[add].n_other++; g.tensors[add].n_fused_add++; }
    op->d2s = d2s;
    g.tensors[out].relu_out = relu != 0;
    op->out_tid = out; op->in_tids = {in, add};
    g.tensors[out].dep_grad_input = g.tensors[in].dep_grad_input || (add >= 0 && g.tensors[add].dep_grad_input);
    return out;
}

int g_chatt(Graph& g, int in, int w1, int b1, int w2, int b2, int Cr, int mode5d_T) {
    const GTensor ti = g.tensors.at(in);
    DL4DS_REQUIRE(g.params.at(w1).n == (size_t)ti.C * Cr && g.params.at(w2).n == (size_t)Cr * ti.C, "chatt: kernel sizes");
    DL4DS_REQUIRE(g.params.at(b1).n == (size_t)Cr && g.params.at(b2).n == (size_t)ti.C, "chatt: bias sizes");
    if (mode5d_T > 0) DL4DS_REQUIRE(ti.nmul == mode5d_T, "chatt: 5-D mode needs nmul == time_window");
    const int out = g.add_tensor(ti.H, ti.W, ti.C, ti.nmul, true, false);
    ChAttOp* op = push<ChAttOp>(g);
    op->in = in; op->out = out; op->w1 = w1; op->b1 = b1; op->w2 = w2; op->b2 = b2; op->Cr = Cr; op->T5 = mode5d_T;
    op->pids = {w1, b1, w2, b2};
    g.tensors[in].n_other++;
    // out = in * scale with scale = sigmoid(..) > 0: behind a ReLU the output is >= 0 and zero exactly where the ReLU's own
    // backward mask is zero, and d(scale) = sum(dout * in) has no term there either -- so a consumer that zeroes d(out) where
    // out <= 0 (a Concatenate's gradient alias, plan_grad_aliases) changes nothing.  The flag only says that much.
    g.tensors[out].relu_out = ti.relu_out;
    op->out_tid = out; op->in_tids = {in};
    g.tensors[out].dep_grad_input = g.tensors[in].dep_grad_input;
    return out;
}

int g_concat(Graph& g, const int* ins, int n) {
    DL4DS_REQUIRE(n >= 2, "concat: need at least two inputs");
    const GTensor t0 = g.tensors.at(ins[0]);
    int C = 0;
    for (int i = 0; i < n; ++i) {
        const GTensor& t = g.tensors.at(ins[i]);
        DL4DS_REQUIRE(t.H == t0.H && t.W == t0.W && t.nmul == t0.nmul,
                      "concat: spatial sizes differ (PadConcat zero-padding is not implemented yet)");
        C += t.C;
    }
    const int out = g.add_tensor(t0.H, t0.W, C, t0.nmul, true, false);
    ConcatOp* op = push<ConcatOp>(g);
    op->ins.assign(ins, ins + n);
    op->out = out;
    for (int i = 0; i < n; ++i) { g.tensors[ins[i]].n_masking++; g.tensors[ins[i]].n_concat_in++; }
    op->out_tid = out; op->in_tids.assign(ins, ins + n);
    for (int i = 0; i < n; ++i) g.tensors[out].dep_grad_input = g.tensors[out].dep_grad_input || g.tensors[ins[i]].dep_grad_input;
    return out;
}

int g_add(Graph& g, int a, int b, int relu) {
    const GTensor ta = g.tensors.at(a), tb = g.tensors.at(b);
    DL4DS_REQUIRE(ta.H == tb.H && ta.W == tb.W && ta.C == tb.C && ta.nmul == tb.nmul, "add: shape mismatch");
    const int out = g.add_tensor(ta.H, ta.W, ta.C, ta.nmul, true, false);
    AddOp* op = push<AddOp>(g);
    op->a = a; op->b = b; op->out = out; op->relu = relu;
    g.tensors[a].n_add_in++;
    g.tensors[b].n_add_in++;
    op->out_tid = out; op->in_tids = {a, b};
    g.tensors[out].dep_grad_input = g.tensors[a].dep_grad_input || g.tensors[b].dep_grad_input;
    return out;
}

int g_act(Graph& g, int in, int kind) {
    const GTensor ti = g.tensors.at(in);
    const int out = g.add_tensor(ti.H, ti.W, ti.C, ti.nmul, true, false);
    ActOp* op = push<ActOp>(g);
    op->in = in; op->out = out; op->act = kind;
    g.tensors[in].n_other++;
    op->out_tid = out; op->in_tids = {in};
    g.tensors[out].dep_grad_input = g.tensors[in].dep_grad_input;
    return out;
}

int g_maxpool2(Graph& g, int in) {
    const GTensor ti = g.tensors.at(in);
    DL4DS_REQUIRE(ti.H >= 2 && ti.W >= 2, "maxpool2: grid too small");
    const int out = g.add_tensor(ti.H / 2, ti.W / 2, ti.C, ti.nmul, true, false);
    MaxPoolOp* op = push<MaxPoolOp>(g);
    op->in = in; op->out = out;
    g.tensors[in].n_masking++;
    g.tensors[in].n_pool_in++;
    op->out_tid = out; op->in_tids = {in};
    g.tensors[out].dep_grad_input = g.tensors[in].dep_grad_input;
    return out;
}

int g_resize(Graph& g, int in, int Ho, int Wo, int nearest) {
    const GTensor ti = g.tensors.at(in);
    const int out = g.add_tensor(Ho, Wo, ti.C, ti.nmul, true, false);
    ResizeOp* op = push<ResizeOp>(g);
    DL4DS_REQUIRE(nearest >= 0 && nearest <= 6, "resize: unknown method");
    op->in = in; op->out = out; op->method = nearest; op->nearest = nearest == 1;
    op->bicubic = nearest >= 2 || (nearest == 0 && !exp_env("DL4DS_RESIZE_BILINEAR_DIRECT"));     // table-driven
    g.tensors[in].n_other++;
    op->out_tid = out; op->in_tids = {in};
    g.tensors[out].dep_grad_input = g.tensors[in].dep_grad_input;
    return out;
}

int g_localconv(Graph& g, int in, int w, int b, int F) {
    const GTensor ti = g.tensors.at(in);
    DL4DS_REQUIRE(g.params.at(w).n == (size_t)ti.H * ti.W * ti.C * F, "localconv: kernel size mismatch");
    if (b >= 0) DL4DS_REQUIRE(g.params.at(b).n == (size_t)ti.H * ti.W * F, "localconv: bias size mismatch");
    const int out = g.add_tensor(ti.H, ti.W, F, ti.nmul, true, false);
    LocalConvOp* op = push<LocalConvOp>(g);
    op->in = in; op->out = out; op->w = w; op->b = b;
    op->pids = {w, b};
    g.tensors[in].n_other++;
    op->out_tid = out; op->in_tids = {in};
    g.tensors[out].dep_grad_input = g.tensors[in].dep_grad_input;
    return out;
}

int g_repeat_time(Graph& g, int in, int T) {
    const GTensor ti = g.tensors.at(in);
    DL4DS_REQUIRE(ti.nmul == 1, "repeat_time: input must be a per-sample tensor");
    const int out = g.add_tensor(ti.H, ti.W, ti.C, T, true, false);
    RepeatTimeOp* op = push<RepeatTimeOp>(g);
    op->in = in; op->out = out; op->T = T;
    g.tensors[in].n_other++;
    g.tensors[out].relu_out = ti.relu_out;       // (T copies of a ReLU output; its gradient is their sum, masked by the producer)
    op->out_tid = out; op->in_tids = {in};
    g.tensors[out].dep_grad_input = g.tensors[in].dep_grad_input;
    return out;
}


// Concatenate without the forward copy (GTensor::alias_of): see graph.h.  Eligible input of a Concatenate: produced by a
// plain Conv2D (no depth_to_space store, no fused attention), a Conv2DTranspose or another Concatenate, read by nothing but
// plain Conv2Ds (as their convolved input), MaxPooling2D / Conv2DTranspose (which read through views) and this one Concatenate,
// not a model input / output.  DenseBlock chains resolve to ONE buffer:
// x_{k+1} = concat(x_k, f(x_k)) makes x_k a channel prefix of x_{k+1}.  DL4DS_NO_CONCAT_ALIAS=1 keeps the copies (A/B, tests).
static void plan_concat_aliases(Graph& g) {
    if (test_env("DL4DS_NO_CONCAT_ALIAS")) return;
    const int nt = (int)g.tensors.size();
    std::vector<int> conv_readers(nt, 0), producer_ok(nt, 0);
    for (auto& up : g.ops) {
        if (ConvOp* c = dynamic_cast<ConvOp*>(up.get())) {
            conv_readers[c->in]++;
            if (c->d2s <= 1 && !c->att_after && !c->att_before) producer_ok[c->out] = 1;
        } else if (ConcatOp* k = dynamic_cast<ConcatOp*>(up.get())) {
            producer_ok[k->out] = 1;
        } else if (up->alias_output() >= 0) {
            // Conv2DTranspose stores through a depth_to_space view; LocalizedConvBlock and the time repeat store through plain
            // views of any channel count / offset (2 = no alignment rule: cfg4's 16 + 8 + 2-channel concatenation)
            const bool any_align = std::string(up->kind) == "localconv" || std::string(up->kind) == "repeat_time";
            producer_ok[up->alias_output()] = (any_align && !exp_env("DL4DS_NO_SLICE_WRITERS")) ? 2 : (any_align ? 0 : 1);
        }
    }
    auto is_output = [&](int t) { for (int o : g.outputs) if (o == t) return true; return false; };
    // in creation order, so that a concatenation that is itself aliased later already knows its own inputs
    for (auto& up : g.ops) {
        ConcatOp* k = dynamic_cast<ConcatOp*>(up.get());
        if (!k) continue;
        int off = 0;
        // a pixel of the concatenation should be whole 32-byte sectors: with 26 channels (104 bytes) no slice is sector-aligned
        // and every producer that stores into it writes partial sectors (cfg4: the 16-channel convolution 519 instead of 302 us,
        // the 8-channel time repeat 394 instead of 69, LocalizedConvBlock 319 instead of 72).  Such a concatenation keeps its own
        // buffer and is written in one pass (concat_join); its inputs may still be concatenations with aligned pixels.
        const bool pitch_ok = (g.tensors[k->out].C & 7) == 0 || exp_env("DL4DS_ALIAS_ANY_PITCH") != nullptr;
        for (int t : k->ins) {
            GTensor& ti = g.tensors[t];
            const bool ok = pitch_ok && producer_ok[t] && !ti.is_input && !is_output(t) && ti.alias_of < 0 && ti.n_add_in == 0 &&
                            ti.n_other == 0 && ti.n_concat_in == 1 && ti.n_masking == 1 + ti.n_pool_in + ti.n_convt_in &&
                            ti.n_conv_in == conv_readers[t] &&
                            (producer_ok[t] == 2 || ((off & 3) == 0 && (ti.C & 3) == 0));
            if (ok) { ti.alias_of = k->out; ti.alias_parent = k->out; ti.alias_coff = off; }
            off += ti.C;
        }
    }
    // resolve chains to their root buffer (an aliased concatenation output forwards its inputs)
    for (int t = 0; t < nt; ++t) {
        GTensor& ti = g.tensors[t];
        if (ti.alias_of < 0) continue;
        int root = ti.alias_of, coff = ti.alias_coff;
        while (g.tensors[root].alias_of >= 0) { coff += g.tensors[root].alias_coff; root = g.tensors[root].alias_of; }
        ti.alias_of = root; ti.alias_coff = coff; ti.alias_ld = g.tensors[root].C;
    }
}


// Concatenate without the backward copies (GTensor::galias).  For a Concatenate K with output R:
//   * every input lives in R's buffer as a direct slice (forward alias, not nested) and R is not itself aliased or a model output;
//   * the inputs' other consumers were created BEFORE K, so in the backward pass K comes first: R's own consumers have written
//     all of dR by then (the first of them without accumulation), the inputs' other consumers accumulate into their slices after;
//   * ReLU masks: an input whose consumers apply its mask (grad_masked) expected K's copy to do so.  Without the copy the mask
//     has to be on dR already -- so if any input is grad_masked, ALL inputs must be ReLU outputs and R becomes grad_masked itself:
//     R's consumers (Conv2D dgrad stores, MaxPooling2D / Conv2DTranspose backward, ...) zero dR where R <= 0, which is each
//     input's own mask on its slice.  (An input that applies its mask itself does it a second time: idempotent.)
// The U-Net decoder levels (PadConcat of the transposed convolution's output and the encoder skip) qualify.
// DL4DS_NO_GRAD_ALIAS=1 keeps the copies (A/B, tests).
static void plan_grad_aliases(Graph& g) {
    if (test_env("DL4DS_NO_GRAD_ALIAS") || test_env("DL4DS_NO_CONCAT_ALIAS")) return;
    auto is_output = [&](int t) { for (int o : g.outputs) if (o == t) return true; return false; };
    const int nops = (int)g.ops.size();
    for (int ik = 0; ik < nops; ++ik) {
        ConcatOp* k = dynamic_cast<ConcatOp*>(g.ops[ik].get());
        if (!k) continue;
        GTensor& tr = g.tensors[k->out];
        if (tr.alias_of >= 0 || !tr.requires_grad || is_output(k->out)) continue;
        bool ok = true, any_masked = false, all_relu = true;
        for (int t : k->ins) {
            const GTensor& ti = g.tensors[t];
            ok = ok && ti.alias_parent == k->out && ti.alias_of == k->out && ti.requires_grad && !ti.is_input;
            // (slices that are not whole channel quads -- LocalizedConvBlock's two channels behind 24 -- keep their dense gradients:
            //  with the 26-channel gradient aliased the 2 -> 24 1x1 dgrad into its 24-channel slice leaves conv_point, +0.64 ms)
            ok = ok && (ti.C & 3) == 0 && (ti.alias_coff & 3) == 0;
            any_masked |= ti.grad_masked;
            all_relu = all_relu && ti.relu_out;
        }
        if (!ok) continue;
        auto reads = [&](GOp* op, int t) {
            if (ConvOp* c = dynamic_cast<ConvOp*>(op)) return c->in == t || c->add == t;
            if (MaxPoolOp* m = dynamic_cast<MaxPoolOp*>(op)) return m->in == t;
            if (ConcatOp* c2 = dynamic_cast<ConcatOp*>(op)) { for (int u : c2->ins) if (u == t) return true; return false; }
            return op->reads_tensor(t);
        };
        for (int j = ik + 1; j < nops && ok; ++j)
            for (int t : k->ins)
                if (reads(g.ops[j].get(), t)) ok = false;
        if (!ok) continue;
        if (any_masked) {
            const bool r_maskable = (tr.n_conv_in + tr.n_add_in + tr.n_masking) >= 1 && tr.n_other == 0 && !exp_env("DL4DS_NO_MASK_FUSION");
            if (!all_relu || !r_maskable) continue;
            tr.grad_masked = true;
            tr.relu_out = true;                      // (every channel of R is a ReLU output)
        }
        for (int t : k->ins) g.tensors[t].galias = true;
    }
}
