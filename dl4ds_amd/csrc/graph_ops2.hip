// Graph ops for the U-Net / CGAN / spatio-temporal configurations: Conv2DTranspose, ConvLSTM2D,
// GlobalAveragePooling, Dense, Dropout.  See graph.h for the runtime contract.
#include "graph.h"
#include "head.h"
#include <algorithm>
#include <cstdlib>

namespace {

inline bool wants_grad(const Graph& g, int tid, const BwdCtx& c) {
    const GTensor& t = g.tensors[tid];
    return t.requires_grad && (!t.is_input || c.input_grads) && (c.param_grads || t.dep_grad_input || exp_env("DL4DS_NO_BWD_PRUNE") != nullptr);
}
template <class T>
T* push(Graph& g) {
    T* p = new T();
    g.ops.emplace_back(p);
    return p;
}
const TView kNone{nullptr, 0, 0, 0, 0, 0, 0, 0};

// ============================================================================================ Conv2DTranspose
struct ConvTOp : GOp {
    int in, w, out, KS, stride, Cout, relu;
    ConvTOp() { kind = "conv2d_transpose"; }
    int alias_output() const override { return out; }
    bool reads_tensor(int t) const override { return t == in; }
    size_t workspace_bytes(Graph& g, int B) override {
        TView x = g.view(in, B, false);
        TView y = g.view(out, B, false);
        return conv2d_transpose_workspace_bytes(x, y, KS, stride);
    }
    void on_finalize(Graph& g) override {
        // ReLU backward of this layer's output folded into its consumers' gradient stores when all of them can (Conv2D dgrad,
        // Add / Concatenate / MaxPooling2D backward, another Conv2DTranspose): same rule as ConvOp
        GTensor& t = g.tensors[out];
        bool is_output = false;
        for (int o : g.outputs) is_output |= (o == out);
        t.grad_masked = relu && !is_output && (t.n_conv_in + t.n_add_in + t.n_masking) >= 1 && t.n_other == 0 &&
                        !exp_env("DL4DS_NO_MASK_FUSION");
    }
    void forward(Graph& g, int B, bool) override {
        conv2d_transpose_forward(g.stream, g.view(in, B, false), g.wp(w), KS, stride, g.view(out, B, false), relu,
                                 g.workspace, g.workspace_bytes);
    }
    void backward(Graph& g, const BwdCtx& c) override {
        if (!g.tensors[out].grad_written) return;
        TView dY = g.view(out, c.B, true, c.b_off, c.b_cnt);
        if (relu && !g.tensors[out].grad_masked)
            bias_act_backward(g.stream, dY, g.view(out, c.B, false, c.b_off, c.b_cnt), dY, nullptr, 0, g.workspace,
                              g.workspace_bytes);
        if (c.param_grads) {
            conv2d_transpose_wgrad(g.stream, g.view(in, c.B, false, c.b_off, c.b_cnt), dY, KS, stride, g.gp(w),
                                   g.params[w].grad_written, g.workspace, g.workspace_bytes);
            g.params[w].grad_written = true;
        }
        if (wants_grad(g, in, c)) {
            TView mask = kNone;
            if (g.tensors[in].grad_masked) mask = g.view(in, c.B, false, c.b_off, c.b_cnt);
            conv2d_transpose_dgrad(g.stream, dY, g.wp(w), KS, stride, g.view(in, c.B, true, c.b_off, c.b_cnt),
                                   g.tensors[in].grad_written, g.workspace, g.workspace_bytes, &mask);
            g.tensors[in].grad_written = true;
        }
    }
};

// ============================================================================================ ConvLSTM2D
struct ConvLSTMOp : GOp {
    int in, out, wk, wr, b, KS, F, T, relu;
    size_t wt_k = 0, wt_r = 0;
    // persistent form (convlstm_seq.hip): one launch per direction for the whole sequence; Z / dZ and the filters it uses
    // keep the gate channels interleaved (column 4 f + gate instead of Keras' gate * F + f)
    bool seq = false;
    size_t wt_kp = 0, wt_up = 0, wt_bp = 0;       // interleaved copies of kernel / recurrent kernel / bias in the Wt scratch
    ConvLSTMOp() { kind = "convlstm2d"; }
    int cin(Graph& g) const { return g.tensors[in].C; }
    void on_finalize(Graph& g) override {
        const GTensor& ti = g.tensors[in];
        seq = convlstm_seq_supported(KS, F, ti.H, ti.W, 1);
        if (seq) {
            wt_kp = g.reserve_wt(g.params[wk].n);
            wt_up = g.reserve_wt(g.params[wr].n);
            wt_bp = g.reserve_wt(g.params[b].n);
            wt_k = g.reserve_wt(g.params[wk].n);
            // dgrad filter of the INTERLEAVED kernel copy (written by every forward pass, read at the start of the backward pass)
            g.add_wt_job(wt_kp, true, wt_k, KS * KS, ti.C, 4 * F);
            return;
        }
        wt_k = g.reserve_wt(g.params[wk].n);
        wt_r = g.reserve_wt(g.params[wr].n);
        g.add_wt_job(g.params[wk].offset, false, wt_k, KS * KS, g.tensors[in].C, 4 * F);     // Graph::refresh_dgrad_weights
        g.add_wt_job(g.params[wr].offset, false, wt_r, KS * KS, F, 4 * F);
    }
    size_t hw(Graph& g) { return (size_t)g.tensors[in].H * g.tensors[in].W; }
    // floats per sample reserved for the recurrence's tile flags (64-bit words, one per 8 x 16 tile of the finest tiling)
    size_t flag_quota(Graph& g) { return 2 * (size_t)cdiv(g.tensors[in].H, 8) * cdiv(g.tensors[in].W, 16) + 64; }
    size_t scratch_floats(Graph& g) {       // dK', dU', db' (interleaved weight gradients) + the tile flags of one launch
        return (size_t)KS * KS * (cin(g) + F) * 4 * F + 4 * F + 64 + flag_quota(g);
    }
    size_t saved_floats_per_sample(Graph& g) override {
        return (size_t)T * hw(g) * (4 * F + F + F + 4 * F) + 2 * hw(g) * F + (seq ? scratch_floats(g) : 0);
    }
    size_t workspace_bytes(Graph& g, int B) override {
        const GTensor& ti = g.tensors[in];
        TView xa = make_view(nullptr, B * T, ti.H, ti.W, ti.C), za = make_view(nullptr, B * T, ti.H, ti.W, 4 * F);
        TView hf = make_view(nullptr, B, ti.H, ti.W, F), zf = make_view(nullptr, B, ti.H, ti.W, 4 * F);
        TView ha = make_view(nullptr, B * T, ti.H, ti.W, F);
        return std::max(conv2d_wgrad_workspace_bytes(xa, za, KS),
                        std::max(conv2d_wgrad_workspace_bytes(hf, zf, KS), conv2d_wgrad_workspace_bytes(ha, za, KS)));
    }
    struct Bufs { float *Z, *C, *H, *dZ, *dh, *dc, *dKp, *dUp, *dbp; unsigned* flags; };
    Bufs bufs(Graph& g, int B) {
        const size_t n = (size_t)B * T * hw(g);
        Bufs r;
        // the tile flags come FIRST and at the size of the slab's largest batch: they are never reset (convlstm_seq.hip:
        // seq_epoch), so no other buffer of any batch size may ever overlay them
        const size_t flag_floats = (flag_quota(g) * g.maxB + 63) & ~(size_t)63;
        r.flags = reinterpret_cast<unsigned*>(saved);
        r.Z = saved + (seq ? flag_floats : 0); r.C = r.Z + n * 4 * F; r.H = r.C + n * F; r.dZ = r.H + n * F;
        r.dh = r.dZ + n * 4 * F; r.dc = r.dh + (size_t)B * hw(g) * F;
        r.dKp = r.dc + (size_t)B * hw(g) * F;
        r.dUp = r.dKp + (size_t)KS * KS * cin(g) * 4 * F;
        r.dbp = r.dUp + (size_t)KS * KS * F * 4 * F;
        return r;
    }
    TView frame(Graph& g, float* base, int B, int t, int ch) {      // frame t of every sample of a (B,T,H,W,ch) buffer
        const GTensor& ti = g.tensors[in];
        TView v = make_view(base + (size_t)t * hw(g) * ch, B, ti.H, ti.W, ch);
        v.nstride = (size_t)T * hw(g) * ch;
        return v;
    }
    void forward(Graph& g, int B, bool) override {
        const GTensor& ti = g.tensors[in];
        Bufs bf = bufs(g, B);
        // H holds the RECURRENT INPUT of every step: H[b, t] = h_{t-1}, H[b, 0] = 0 (zero initial state).  Stored that way
        // the recurrent kernel's weight gradient sum_{b,t} wgrad(h_{t-1}, dZ_t) is ONE convolution-wgrad over the B*T frame
        // pairs (H, dZ) -- frame 0 of every sample contributes nothing -- instead of T-1 launches of ~60 us.
        // (the one-launch recurrence writes that zero frame itself)
        if (!seq) HIP_CHECK(hipMemset2DAsync(bf.H, (size_t)T * hw(g) * F * sizeof(float), 0, hw(g) * F * sizeof(float), (size_t)B, g.stream));
        if (seq) {
            DL4DS_REQUIRE(convlstm_seq_flag_bytes(ti.H, ti.W, B) <= flag_quota(g) * sizeof(float) * (size_t)B && B <= g.maxB, "convlstm: flag area");
            float *Kp = g.Wt + wt_kp, *Up = g.Wt + wt_up, *bp = g.Wt + wt_bp;
            {   // kernel, recurrent kernel and bias into the interleaved gate order: one launch
                const float* src[3] = {g.wp(wk), g.wp(wr), g.wp(b)};
                float* dst[3] = {Kp, Up, bp};
                const int rows[3] = {KS * KS * ti.C, KS * KS * F, 1}, acc[3] = {0, 0, 0};
                convlstm_gate_interleave_n(g.stream, 3, src, dst, rows, acc, F, true);
            }
            ConvEpilogue ep;
            ep.bias = bp;
            conv2d_forward(g.stream, g.view(in, B, false), Kp, KS, make_view(bf.Z, B * T, ti.H, ti.W, 4 * F), ep);
            convlstm_seq_forward(g.stream, Up, bf.Z, bf.C, bf.H, g.tensors[out].data, bf.flags, B, T, ti.H, ti.W, KS, F, relu);
            return;
        }
        ConvEpilogue ep;
        ep.bias = g.wp(b);
        conv2d_forward(g.stream, g.view(in, B, false), g.wp(wk), KS, make_view(bf.Z, B * T, ti.H, ti.W, 4 * F), ep);
        for (int t = 0; t < T; ++t) {
            TView zt = frame(g, bf.Z, B, t, 4 * F);
            if (t > 0) {
                ConvEpilogue er;
                er.accumulate = 1;
                conv2d_forward(g.stream, frame(g, bf.H, B, t, F), g.wp(wr), KS, zt, er);
            }
            TView hnext = frame(g, bf.H, B, std::min(t + 1, T - 1), F);
            if (t + 1 >= T) hnext.p = nullptr;
            convlstm_gates_forward(g.stream, zt, frame(g, bf.C, B, t > 0 ? t - 1 : 0, F), frame(g, bf.C, B, t, F), hnext,
                                   frame(g, g.tensors[out].data, B, t, F), relu, t == 0);
        }
    }
    bool partial_batch_ok() const override { return false; }
    void backward(Graph& g, const BwdCtx& c) override {
        if (!g.tensors[out].grad_written) return;
        DL4DS_REQUIRE(c.b_off == 0 && (c.b_cnt < 0 || c.b_cnt == c.B), "convlstm: partial-batch backward not supported");
        const int B = c.B;
        const GTensor& ti = g.tensors[in];
        Bufs bf = bufs(g, B);
        TView dZall = make_view(bf.dZ, B * T, ti.H, ti.W, 4 * F);
        if (seq) {
            convlstm_seq_backward(g.stream, g.Wt + wt_up, bf.Z, bf.C, g.tensors[out].data, g.tensors[out].grad, bf.dZ, bf.dc, bf.flags,
                                  B, T, ti.H, ti.W, KS, F, relu);
            if (c.param_grads) {
                // weight gradients against the interleaved dZ come out with interleaved columns: back to Keras' order on the
                // way into the gradient arena
                conv2d_wgrad(g.stream, g.view(in, B, false), dZall, KS, bf.dKp, 0, bf.dbp, 0, g.workspace, g.workspace_bytes);
                if (T > 1)
                    conv2d_wgrad(g.stream, make_view(bf.H, B * T, ti.H, ti.W, F), dZall, KS, bf.dUp, 0, nullptr, 0, g.workspace,
                                 g.workspace_bytes);
                {   // (one launch for the two or three arrays)
                    const float* src[3] = {bf.dKp, bf.dbp, bf.dUp};
                    float* dst[3] = {g.gp(wk), g.gp(b), g.gp(wr)};
                    const int rows[3] = {KS * KS * ti.C, 1, KS * KS * F};
                    const int acc[3] = {(int)g.params[wk].grad_written, (int)g.params[b].grad_written, (int)g.params[wr].grad_written};
                    convlstm_gate_interleave_n(g.stream, T > 1 ? 3 : 2, src, dst, rows, acc, F, false);
                }
                g.params[wk].grad_written = g.params[b].grad_written = true;
                if (T > 1) g.params[wr].grad_written = true;
            }
            if (wants_grad(g, in, c)) {
                ConvEpilogue ep;
                ep.accumulate = g.tensors[in].grad_written;
                conv2d_forward(g.stream, dZall, g.Wt + wt_k, KS, g.view(in, B, true), ep);
                g.tensors[in].grad_written = true;
            }
            return;
        }
        float* Ut = g.Wt + wt_r;
        TView dh = make_view(bf.dh, B, ti.H, ti.W, F), dc = make_view(bf.dc, B, ti.H, ti.W, F);
        for (int t = T - 1; t >= 0; --t) {
            TView dzt = frame(g, bf.dZ, B, t, 4 * F);
            convlstm_gates_backward(g.stream, frame(g, bf.Z, B, t, 4 * F), frame(g, bf.C, B, t > 0 ? t - 1 : 0, F),
                                    frame(g, bf.C, B, t, F), frame(g, g.tensors[out].data, B, t, F),
                                    frame(g, g.tensors[out].grad, B, t, F), dh, dc, dzt, relu, t == 0, t == T - 1);
            if (t > 0) {
                ConvEpilogue ep;
                conv2d_forward(g.stream, dzt, Ut, KS, dh, ep);
            }
        }
        if (c.param_grads) {
            conv2d_wgrad(g.stream, g.view(in, B, false), dZall, KS, g.gp(wk), g.params[wk].grad_written, g.gp(b),
                         g.params[b].grad_written, g.workspace, g.workspace_bytes);
            g.params[wk].grad_written = g.params[b].grad_written = true;
            // recurrent kernel: sum over (sample, t) of wgrad(h_{t-1}, dZ_t) = one wgrad over the B*T frame pairs (H, dZ)
            // (H[b, t] = h_{t-1}, H[b, 0] = 0: see forward)
            if (T > 1) {
                conv2d_wgrad(g.stream, make_view(bf.H, B * T, ti.H, ti.W, F), dZall, KS, g.gp(wr), g.params[wr].grad_written,
                             nullptr, 0, g.workspace, g.workspace_bytes);
                g.params[wr].grad_written = true;
            }
        }
        if (wants_grad(g, in, c)) {
            float* Kt = g.Wt + wt_k;
            ConvEpilogue ep;
            ep.accumulate = g.tensors[in].grad_written;
            conv2d_forward(g.stream, dZall, Kt, KS, g.view(in, B, true), ep);
            g.tensors[in].grad_written = true;
        }
    }
};

// ============================================================================================ GlobalAveragePooling2D
struct GapOp : GOp {
    int in, out;
    bool over_time = false;    // GlobalAveragePooling3D: the frames of a sample are averaged too (discriminator.py:74)
    GapOp() { kind = "gap"; }
    size_t workspace_bytes(Graph& g, int B) override { return gap_workspace_bytes(B * g.tensors[in].nmul, g.tensors[in].C); }
    void forward(Graph& g, int B, bool) override {
        const GTensor& ti = g.tensors[in];
        const int fr = over_time ? ti.nmul : 1;
        gap_forward(g.stream, ti.data, g.tensors[out].data, B * ti.nmul / fr, ti.H * ti.W * fr, ti.C, g.workspace,
                    g.workspace_bytes);
    }
    void backward(Graph& g, const BwdCtx& c) override {
        if (!g.tensors[out].grad_written || !wants_grad(g, in, c)) return;
        const GTensor& ti = g.tensors[in];
        const int fr = over_time ? ti.nmul : 1;
        const int cnt = (c.b_cnt < 0 ? c.B : c.b_cnt) * ti.nmul / fr;
        // (a ReLU output whose consumers apply its mask: this store does -- the tensor is dense, pooling never reads an alias)
        const float* mask = (ti.grad_masked && ti.alias_of < 0) ? ti.data + (size_t)c.b_off * ti.per_sample() : nullptr;
        DL4DS_REQUIRE(!ti.grad_masked || mask, "gap: masked gradient of an aliased tensor");
        gap_backward(g.stream, g.tensors[out].grad + (size_t)c.b_off * (ti.nmul / fr) * ti.C,
                     ti.grad + (size_t)c.b_off * ti.per_sample(), cnt, ti.H * ti.W * fr, ti.C, ti.grad_written, mask);
        g.tensors[in].grad_written = true;
    }
};

// ============================================================================================ Dense (+activation)
struct DenseOp : GOp {
    int in, out, w, b, F, act;
    DenseOp() { kind = "dense"; }
    void forward(Graph& g, int B, bool) override {
        const GTensor& ti = g.tensors[in];
        dense_forward(g.stream, ti.data, g.wp(w), b >= 0 ? g.wp(b) : nullptr, g.tensors[out].data, B * ti.nmul, ti.C, F, act);
    }
    void backward(Graph& g, const BwdCtx& c) override {
        if (!g.tensors[out].grad_written) return;
        const GTensor& ti = g.tensors[in];
        const int cnt = (c.b_cnt < 0 ? c.B : c.b_cnt) * ti.nmul;
        const bool dx = wants_grad(g, in, c);
        dense_backward(g.stream, ti.data, g.wp(w), g.tensors[out].data, g.tensors[out].grad, dx ? ti.grad : nullptr,
                       ti.grad_written, g.gp(w), b >= 0 ? g.gp(b) : nullptr, g.params[w].grad_written, c.param_grads,
                       c.b_off * ti.nmul, cnt, ti.C, F, act);
        if (dx) g.tensors[in].grad_written = true;
        if (c.param_grads) {
            g.params[w].grad_written = true;
            if (b >= 0) g.params[b].grad_written = true;
        }
    }
};

// ============================================================================================ Dropout
struct DropoutOp : GOp {
    int in, out;
    float rate;
    int variant = 0;           // 0 Dropout, 1 GaussianDropout, 2 SpatialDropout2D/3D (blocks.py:679-701)
    int spatial_dim = 2;       // SpatialDropout3D shares the channel mask over the time axis too
    bool mc = false;           // MC* layers stay active at inference (blocks.py:658-676)
    bool injected = false;
    unsigned long long counter = 0;
    DropoutOp() { kind = "dropout"; }
    // mask entries per sample: one per element, or one per (frame, channel) / (sample, channel) for the spatial variants
    size_t mask_per_sample(Graph& g) const {
        const GTensor& t = g.tensors[in];
        if (variant != 2) return t.per_sample();
        return (size_t)(spatial_dim == 3 ? 1 : t.nmul) * t.C;
    }
    size_t inner(Graph& g) const {
        const GTensor& t = g.tensors[in];
        return (size_t)t.H * t.W * (spatial_dim == 3 ? t.nmul : 1);
    }
    float scale() const { return variant == 1 ? 1.f : 1.f / (1.f - rate); }
    size_t saved_floats_per_sample(Graph& g) override { return mask_per_sample(g); }
    size_t mask_floats(Graph& g, int B) override { return mask_per_sample(g) * B; }
    bool set_mask(Graph& g, const float* host, size_t n) override {
        HIP_CHECK(hipMemcpyAsync(saved, host, n * sizeof(float), hipMemcpyHostToDevice, g.stream));
        injected = true;
        return true;
    }
    void apply(Graph& g, const float* x, const float* mask, float* y, size_t n, int acc) {
        if (variant == 2) dropout_apply_bcast(g.stream, x, mask, y, n, scale(), acc, g.tensors[in].C, inner(g));
        else dropout_apply(g.stream, x, mask, y, n, scale(), acc);
    }
    void forward(Graph& g, int B, bool training) override {
        const size_t n = g.tensors[in].per_sample() * B;
        if (!training && !mc) {
            HIP_CHECK(hipMemcpyAsync(g.tensors[out].data, g.tensors[in].data, n * sizeof(float), hipMemcpyDeviceToDevice,
                                     g.stream));
            return;
        }
        if (!injected)
            dropout_make_mask(g.stream, saved, mask_per_sample(g) * B, rate, seed + (++counter) * 0x1000003ull, variant == 1);
        injected = false;
        apply(g, g.tensors[in].data, saved, g.tensors[out].data, n, 0);
    }
    void backward(Graph& g, const BwdCtx& c) override {
        if (!g.tensors[out].grad_written || !wants_grad(g, in, c)) return;
        const size_t ps = g.tensors[in].per_sample();
        const size_t off = (size_t)c.b_off * ps, n = (size_t)(c.b_cnt < 0 ? c.B : c.b_cnt) * ps;
        apply(g, g.tensors[out].grad + off, saved + (size_t)c.b_off * mask_per_sample(g), g.tensors[in].grad + off, n,
              g.tensors[in].grad_written);
        g.tensors[in].grad_written = true;
    }
    unsigned long long seed = 0x5DEECE66Dull;
};

}  // namespace

int g_conv2d_transpose(Graph& g, int in, int w, int KS, int stride, int Cout, int relu) {
    const GTensor ti = g.tensors.at(in);
    DL4DS_REQUIRE(g.params.at(w).n == (size_t)KS * KS * Cout * ti.C, "conv2d_transpose: kernel parameter size mismatch");
    const int out = g.add_tensor(ti.H * stride, ti.W * stride, Cout, ti.nmul, true, false);
    ConvTOp* op = push<ConvTOp>(g);
    op->in = in; op->w = w; op->out = out; op->KS = KS; op->stride = stride; op->Cout = Cout; op->relu = relu;
    g.tensors[in].n_masking++;              // (its dgrad store applies the producer's ReLU mask, like a Conv2D consumer)
    g.tensors[in].n_convt_in++;
    g.tensors[out].relu_out = relu != 0;
    op->pids = {w};
    op->out_tid = out; op->in_tids = {in};
    g.tensors[out].dep_grad_input = g.tensors[in].dep_grad_input;
    return out;
}

int g_convlstm(Graph& g, int in, int wk, int wr, int b, int KS, int F, int T, int relu) {
    const GTensor ti = g.tensors.at(in);
    DL4DS_REQUIRE(ti.nmul == T, "convlstm: input tensor must carry the time window as batch multiplier");
    DL4DS_REQUIRE(KS == 1 || KS == 3 || KS == 5, "convlstm: kernel size must be 1, 3 or 5");
    DL4DS_REQUIRE(g.params.at(wk).n == (size_t)KS * KS * ti.C * 4 * F, "convlstm: kernel size mismatch");
    DL4DS_REQUIRE(g.params.at(wr).n == (size_t)KS * KS * F * 4 * F, "convlstm: recurrent kernel size mismatch");
    DL4DS_REQUIRE(g.params.at(b).n == (size_t)4 * F, "convlstm: bias size mismatch");
    const int out = g.add_tensor(ti.H, ti.W, F, T, true, false);
    ConvLSTMOp* op = push<ConvLSTMOp>(g);
    op->in = in; op->out = out; op->wk = wk; op->wr = wr; op->b = b; op->KS = KS; op->F = F; op->T = T; op->relu = relu;
    g.tensors[in].n_other++;
    op->pids = {wk, wr, b};
    op->out_tid = out; op->in_tids = {in};
    g.tensors[out].dep_grad_input = g.tensors[in].dep_grad_input;
    return out;
}

int g_gap(Graph& g, int in, int over_time) {
    const GTensor ti = g.tensors.at(in);
    const int out = g.add_tensor(1, 1, ti.C, over_time ? 1 : ti.nmul, true, false);
    GapOp* op = push<GapOp>(g);
    op->in = in; op->out = out; op->over_time = over_time != 0;
    // a masking consumer (like MaxPooling2D / Concatenate): its backward applies the input's ReLU mask (DL4DS_NO_GAP_MASK=1: as before)
    if (exp_env("DL4DS_NO_GAP_MASK")) g.tensors[in].n_other++; else g.tensors[in].n_masking++;
    op->out_tid = out; op->in_tids = {in};
    g.tensors[out].dep_grad_input = g.tensors[in].dep_grad_input;
    return out;
}

int g_dense(Graph& g, int in, int w, int b, int F, int act) {
    const GTensor ti = g.tensors.at(in);
    DL4DS_REQUIRE(ti.H == 1 && ti.W == 1, "dense: input must be (B,1,1,C)");
    DL4DS_REQUIRE(g.params.at(w).n == (size_t)ti.C * F, "dense: kernel size mismatch");
    DL4DS_REQUIRE(act == ACT_NONE || act == ACT_SIGMOID || act == ACT_RELU || act == ACT_TANH, "dense: activation");
    const int out = g.add_tensor(1, 1, F, ti.nmul, true, false);
    DenseOp* op = push<DenseOp>(g);
    op->in = in; op->out = out; op->w = w; op->b = b; op->F = F; op->act = act;
    g.tensors[in].n_other++;
    op->pids = {w, b};
    op->out_tid = out; op->in_tids = {in};
    g.tensors[out].dep_grad_input = g.tensors[in].dep_grad_input;
    return out;
}

int g_dropout(Graph& g, int in, float rate, int variant, int mc, int spatial_dim) {
    const GTensor ti = g.tensors.at(in);
    DL4DS_REQUIRE(rate >= 0.f && rate < 1.f, "dropout: rate must be in [0,1)");
    DL4DS_REQUIRE(variant >= 0 && variant <= 2 && (spatial_dim == 2 || spatial_dim == 3), "dropout: bad variant");
    const int out = g.add_tensor(ti.H, ti.W, ti.C, ti.nmul, true, false);
    DropoutOp* op = push<DropoutOp>(g);
    op->in = in; op->out = out; op->rate = rate; op->variant = variant; op->mc = mc != 0; op->spatial_dim = spatial_dim;
    op->seed += 0x9E3779B97F4A7C15ull * (g.dropout_ops.size() + 1);
    g.tensors[in].n_other++;
    g.dropout_ops.push_back(op);
    op->out_tid = out; op->in_tids = {in};
    g.tensors[out].dep_grad_input = g.tensors[in].dep_grad_input;
    return out;
}
