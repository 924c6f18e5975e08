// Static op-list runtime for the dl4ds conv-SR train step on one MI355X.
//
// The Python builders (dl4ds_amd/models/*.py -- same signatures as dl4ds.models.*) lower a model into
// this graph through the C ABI: tensors (NHWC, batch = B * nmul), parameters (slices of ONE flat fp32
// arena, so Adam and the RCCL all-reduce each touch a single buffer), and ops with hand-written
// forward/backward.  No autograd tape, no allocator calls inside a step, one HIP stream.
#pragma once
#include "ops.h"
#include <memory>
#include <string>
#include <vector>

struct Graph;

struct GTensor {
    int H = 0, W = 0, C = 0;
    int nmul = 1;              // batch multiplier: N = B * nmul (time_window for TimeDistributed tensors)
    bool requires_grad = true;
    bool is_input = false;
    bool dep_grad_input = false;   // depends (transitively) on an input tensor that takes a gradient (set by the op constructors)
    float* data = nullptr;
    float* grad = nullptr;
    bool grad_written = false;
    // consumers: as the convolved input of a Conv2D / as anything else (residual operand, concat, pooling, ...)
    int n_conv_in = 0, n_other = 0;
    int n_add_in = 0;          // ... as an operand of an Add (which can apply the ReLU mask while copying its gradient)
    int n_masking = 0;         // ... as an input of a Concatenate / MaxPooling2D (their backward applies the mask, too)
    int n_concat_in = 0;       // ... of which Concatenates
    int n_fused_add = 0;       // ... (counted in n_other) as the residual operand fused into a Conv2D's epilogue
    bool two_add_inplace = false;   // feeds ONE Conv2D and TWO fused adds, planned copy-free (ConvOp::on_prepare, round 5)
    int n_pool_in = 0, n_convt_in = 0;   // ... (counted in n_masking) as the input of a MaxPooling2D / Conv2DTranspose
    bool relu_out = false;     // written by a layer whose fused activation is ReLU
    // residual-block input during a backward pass: the gradient arriving through the block's fused add (the add's dZ) that
    // the block's first convolution will fold into its dgrad store instead of a copy + an accumulating store (null: none)
    const float* pending_add = nullptr;
    TView pending_view{};      // the pending operand with ITS layout (it may be a channel slice of a Concatenate's gradient: pitch != C)
    // Concatenate without the forward copy: the activation lives INSIDE the concatenation's buffer (channel offset
    // alias_coff of tensor alias_of, pixel pitch alias_ld = that buffer's channel count); decided at finalize for tensors
    // written by a Conv2D / Concatenate and read only by Conv2Ds and ONE Concatenate.  Gradients stay dense.
    int alias_of = -1, alias_coff = 0, alias_ld = 0;
    int alias_parent = -1;     // the Concatenate output it is a direct input slice of (alias_of = the root of a chain of them)
    // ... and the gradient too (round 2): d(loss)/d(this tensor) is the same channel slice of the concatenation's gradient
    // buffer, so Concatenate's backward copies nothing for it; its other consumers accumulate into the slice, its producer
    // reads its dZ from it (plan_grad_aliases in graph.hip lists the conditions)
    bool galias = false;
    // ReLU backward fused into the writers: every consumer is a Conv2D (or an Add), whose dgrad epilogue zeroes the gradient
    // where this activation is <= 0 (the mask is linear, so each accumulating writer applies it independently)
    bool grad_masked = false;
    size_t per_sample() const { return (size_t)nmul * H * W * C; }
};

struct GParam {
    size_t offset = 0, n = 0;
    bool grad_written = false;
};

struct BwdCtx {
    int B;                 // batch of the forward pass the saved activations belong to
    int b_off, b_cnt;      // sample range to back-propagate (CGAN: the "fake" half of a 2B batch)
    bool param_grads;      // compute parameter gradients
    bool input_grads;      // propagate into tensors flagged is_input (CGAN generator path)
};

struct GOp {
    virtual ~GOp() {}
    virtual void forward(Graph& g, int B, bool training) = 0;
    virtual void backward(Graph& g, const BwdCtx& c) = 0;
    virtual size_t workspace_bytes(Graph& g, int B) { return 0; }
    virtual size_t saved_floats_per_sample(Graph& g) { return 0; }   // op-private saved activations
    float* saved = nullptr;
    std::vector<int> pids;     // parameters this op reads (set by the op constructors; drives gradient bucketing)
    std::vector<int> in_tids;  // tensors this op reads (negative entries: absent optional operands) and the one it writes
    int out_tid = -1;
    virtual void on_finalize(Graph& g) {}
    virtual void on_prepare(Graph& g) {}      // after (re)allocation of the activation / gradient buffers
    virtual void on_resolve_aliases(Graph& g) {}   // after every op's on_prepare, in reverse op order: gradient buffers shared along chains
    virtual bool partial_batch_ok() const { return true; }   // backward over a sample sub-range (BwdCtx::b_off / b_cnt)
    // ops with batch statistics: the forward batch is `groups` independent sub-batches (CGAN: [real ; fake] = the reference's
    // two discriminator calls), each normalised by its own statistics, moving averages updated group after group
    virtual void set_batch_groups(int groups) {}
    virtual bool set_mask(Graph& g, const float* host, size_t n) { return false; }   // dropout keep-mask injection
    virtual size_t mask_floats(Graph& g, int B) { return 0; }                         // size of the mask of the last forward
    const char* kind = "op";
    virtual std::string describe_fusion(Graph& g) { return ""; }
    // >= 0: the tensor this op writes THROUGH A VIEW (so it may live inside a Concatenate's buffer, GTensor::alias_of)
    virtual int alias_output() const { return -1; }
    // does this op read tensor t as an input?  (only the op kinds that may read an aliased tensor have to answer: plan_grad_aliases)
    virtual bool reads_tensor(int t) const { return false; }    // non-empty: JSON object describing what this op handed to its neighbours
};

struct Graph {
    std::vector<GTensor> tensors;
    std::vector<GParam> params;
    std::vector<std::unique_ptr<GOp>> ops;
    std::vector<int> inputs, outputs;
    std::vector<GOp*> dropout_ops;
    size_t n_params = 0;
    float* W = nullptr;        // parameter arena
    float* G = nullptr;        // gradient arena
    float* Wt = nullptr;       // scratch arena (flipped/transposed or re-arranged weights for dgrad/deconv)
    size_t wt_floats = 0;
    bool own_arena = true;
    int maxB = 0;
    bool finalized = false;
    float* workspace = nullptr;
    size_t workspace_bytes = 0;
    // weight-gradient kernels run on a second stream (own workspace) concurrently with the dgrad chain:
    // the two MFMA streams fill each other's staging / epilogue bubbles on the CUs
    hipStream_t aux_stream = nullptr;
    float* aux_workspace = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool aux_used = false;
    void fork_aux();            // aux_stream waits for everything enqueued on `stream` so far
    void join_aux();            // `stream` waits for everything enqueued on aux_stream
    hipStream_t stream = nullptr;
    std::vector<float*> allocations;
    // dgrad filter arrangements (flipped taps, cin <-> cout) of every convolution: registered by the ops at finalize
    // and produced by ONE launch at the start of the first backward pass after a forward (the filters only change in
    // the optimiser step), instead of one tiny launch per layer
    struct WtJob { size_t src_off; bool src_in_wt; size_t dst_off; int KK, Cin, Cout; };
    std::vector<WtJob> wt_jobs;
    void* wt_jobs_dev = nullptr;
    int wt_job_blocks = 0;
    bool wt_fresh = false;
    void add_wt_job(size_t src_off, bool src_in_wt, size_t dst_off, int KK, int Cin, int Cout) {
        wt_jobs.push_back({src_off, src_in_wt, dst_off, KK, Cin, Cout});
    }
    void refresh_dgrad_weights();
    // Gradient buckets for the data-parallel all-reduce (contiguous arena ranges in creation = forward order).
    // Bucket k is final once the backward pass has run op `ready_op` (the first forward op using any of its
    // parameters); Graph::backward then calls grad_ready(offset, count) -- the trainer launches that bucket's
    // ncclAllReduce on the communication stream while the rest of the backward pass is still running.
    struct GradBucket { size_t off = 0, n = 0; int p_lo = 0, p_hi = 0; int ready_op = -1; };
    std::vector<GradBucket> buckets;
    // `aux` (may be null): second stream that also wrote into the bucket (weight gradients); the collective must
    // wait for both, the compute streams wait for neither
    void (*grad_ready)(void* ctx, float* grads, size_t n, hipStream_t stream, hipStream_t aux) = nullptr;
    void* grad_ready_ctx = nullptr;
    void plan_buckets(size_t target_bytes);
    // Shared sub-graph of a batch that is `groups` copies of the SAME samples on some inputs (CGAN: the discriminator sees
    // [real ; fake], and its conditioning branch gets the same array in both halves -- the reference's two discriminator calls
    // evaluate that branch twice on identical data, cgan.py:598-599).  Ops whose inputs all descend from such inputs run on the
    // first B / groups samples only; a tensor they hand to the rest of the graph is copied to the other groups in forward, its
    // gradient groups are summed into the first before the shared producer's backward (the branch is linear in its output
    // gradient and its ReLU masks are the same in every group, so back-propagating the sum once is the sum of the groups'
    // parameter gradients).  plan_shared() decides; shared_groups > 1 switches it on for the forward / backward calls that follow.
    std::vector<char> op_shared, t_boundary;
    int shared_groups = 1;
    bool plan_shared(const std::vector<int>& dup_inputs);

    ~Graph();
    int add_tensor(int H, int W, int C, int nmul, bool requires_grad, bool is_input);
    int add_param(size_t n);
    size_t reserve_wt(size_t floats) { size_t o = wt_floats; wt_floats += (floats + 3) & ~(size_t)3; return o; }
    void finalize();
    void prepare(int B);                    // (re)allocate activations / workspace for batch B
    void forward(int B, bool training);
    void zero_grad_flags();
    void backward(const BwdCtx& c);
    TView view(int tid, int B, bool grad, int b_off = 0, int b_cnt = -1) const;
    float* wp(int pid) const { return W + params[pid].offset; }
    float* gp(int pid) const { return G + params[pid].offset; }
};

// ---- op constructors (graph.hip)
int g_conv2d(Graph& g, int in, int w, int b, int add, int KS, int Cout, int relu, int d2s);
int g_conv2d_transpose(Graph& g, int in, int w, int KS, int stride, int Cout, int relu);
int g_chatt(Graph& g, int in, int w1, int b1, int w2, int b2, int Cr, int mode5d_T);
int g_concat(Graph& g, const int* ins, int n);
int g_add(Graph& g, int a, int b, int relu);
int g_act(Graph& g, int in, int kind);
int g_maxpool2(Graph& g, int in);
int g_resize(Graph& g, int in, int Ho, int Wo, int nearest = 0);
int g_localconv(Graph& g, int in, int w, int b, int F);
int g_repeat_time(Graph& g, int in, int T);
int g_convlstm(Graph& g, int in, int wk, int wr, int b, int KS, int F, int T, int relu);
int g_gap(Graph& g, int in, int T);
int g_dense(Graph& g, int in, int w, int b, int F, int act);
int g_dropout(Graph& g, int in, float rate, int variant = 0, int mc = 0, int spatial_dim = 2);
int g_conv2d_folded(Graph& g, int in, int w1, int b1, int w2, int b2, int KS, int Cmid, int Cout, int relu, int d2s, int aux = -1);
int g_pad(Graph& g, int in, int Ho, int Wo);
int g_dwconv(Graph& g, int in, int w, int b, int KS);
int g_slice(Graph& g, int in, int oy, int ox, int step, int Ho, int Wo);
int g_norm(Graph& g, int in, int gamma, int beta, int mov_mean, int mov_var, int batch, float eps, int relu);
// the recurrent nets' tail [x, repeat(s), LocalizedConvBlock] -> TransitionLast as one op (graph_ops4.hip)
bool rec_tail_supported(int CX, int CS, int CO);
int g_rec_tail(Graph& g, int x, int s, int wt, int bt, int wl, int bl, int w, int b, int T, int CO);

// ---- trainer (trainer.hip)
struct AdamCfg {
    float lr0 = 1e-3f, lr1 = 1e-3f;
    double boundary = 1e30;        // PiecewiseConstantDecay([boundary],[lr0,lr1]) -- supervised.py:340-346
    float beta1 = 0.9f, beta2 = 0.999f, eps = 1e-7f;
};

struct Trainer {
    Graph* g = nullptr;
    AdamCfg cfg;
    float* m = nullptr;
    float* v = nullptr;
    long step = 0;                 // optimizer.iterations
    int loss_kind = LOSS_MAE;
    float* d_loss = nullptr;       // device scalars [8]
    float* loss_ws = nullptr;
    size_t loss_ws_bytes = 0;
    float* y_true = nullptr;
    size_t y_true_floats = 0;
    ~Trainer();
};
