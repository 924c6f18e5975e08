// extern "C" boundary of libdl4ds_hip.so -- see include/dl4ds_hip.h for the contract of every symbol.
#include "../../include/dl4ds_hip.h"
#include "graph.h"
#include "runtime.h"
#include "dist.h"
#include "prof.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <mutex>
#include <vector>

// ---- internal functions defined in other translation units
Trainer* trainer_create(Graph* g, int loss_kind, const AdamCfg& cfg);
void graph_load_inputs(Graph& g, const float* const* inputs, int n_inputs, int B, bool is_host);
void trainer_evaluate(Trainer& t, const float* const* inputs, int n_inputs, const float* y_true, int B, bool is_host);
void trainer_loss_and_grads(Trainer& t, const float* const* inputs, int n_inputs, const float* y_true, int B, bool is_host,
                            bool reduce_across_ranks);
void trainer_step(Trainer& t, const float* const* inputs, int n_inputs, const float* y_true, int B, bool is_host, float* loss_host);
struct CganTrainer;
CganTrainer* cgan_create(Graph* gen, Graph* disc, int px_loss_kind, float lr, float beta1, float lam);
void cgan_destroy(CganTrainer* t);
void cgan_step(CganTrainer& t, const float* const* gen_inputs, int n_gen_inputs, const float* hr, int B, bool is_host,
               const float* dropout_keep_host, bool apply_update, float* losses_host);
Trainer* cgan_disc_trainer(CganTrainer* t);
Trainer* cgan_gen_trainer(CganTrainer* t);
void cgan_set_learning_rates(CganTrainer* t, float gen_lr, float disc_lr);

namespace {
thread_local std::string g_last_error;
Runtime g_rt;
std::mutex g_rt_mu;
}  // namespace

Runtime& rt() { return g_rt; }
void set_last_error(const std::string& s) { g_last_error = s; }

// ---- sticky device-side error word (runtime.h)
namespace {
unsigned* g_err_host = nullptr;       // pinned, mapped
unsigned* g_err_dev = nullptr;
}  // namespace
unsigned* device_error_word() {
    if (!g_err_dev) {
        HIP_CHECK(hipHostMalloc((void**)&g_err_host, sizeof(unsigned), hipHostMallocMapped));
        *g_err_host = 0u;
        HIP_CHECK(hipHostGetDevicePointer((void**)&g_err_dev, g_err_host, 0));
    }
    return g_err_dev;
}
unsigned* device_error_word_if_any() { return g_err_dev; }
void device_error_check(const char* what) {
    if (!g_err_host) return;
    const unsigned code = *reinterpret_cast<volatile unsigned*>(g_err_host);
    if (!code) return;
    *g_err_host = 0u;                  // reported once; the step that raised it is lost either way
    std::string msg = std::string(what) + ": a kernel reported a device-side error (code " + std::to_string(code) + ")";
    if (code & DEV_ERR_CONVLSTM_SEQ_TIMEOUT)
        msg += ": the persistent ConvLSTM kernel gave up waiting for a neighbouring tile (its workgroups were not all resident, "
               "e.g. behind kernels of another stream or an RCCL collective held up by a late rank); the activations and "
               "gradients of that step are INVALID.  DL4DS_NO_CONVLSTM_SEQ=1 runs the recurrence step by step";
    throw Dl4dsError(msg);
}

void rt_ensure_init() {
    std::lock_guard<std::mutex> lk(g_rt_mu);
    if (g_rt.inited) return;
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    g_rt.device = dev;
    HIP_CHECK(hipStreamCreateWithFlags(&g_rt.stream, hipStreamNonBlocking));
    HIP_CHECK(hipStreamCreateWithFlags(&g_rt.comm_stream, hipStreamNonBlocking));
    HIP_CHECK(hipStreamCreateWithFlags(&g_rt.aux_stream, hipStreamNonBlocking));
    HIP_CHECK(hipEventCreate(&g_rt.ev0));
    HIP_CHECK(hipEventCreate(&g_rt.ev1));
    g_rt.inited = true;
}

#define API_BEGIN try {
#define API_END                                   \
    }                                             \
    catch (const std::exception& e) {             \
        set_last_error(e.what());                 \
        return -1;                                \
    }                                             \
    catch (...) {                                 \
        set_last_error("unknown C++ exception");  \
        return -1;                                \
    }                                             \
    return 0;

struct dl4ds_graph { Graph g; };
struct dl4ds_trainer {
    Trainer* t = nullptr;          // supervised trainer
    CganTrainer* c = nullptr;      // or a CGAN trainer
};

static hipStream_t S() { rt_ensure_init(); return rt().stream; }
static float* scratch(size_t bytes) {       // grow-only scratch for the single-op entry points
    static float* p = nullptr;
    static size_t cap = 0;
    if (bytes > cap) {
        HIP_CHECK(hipStreamSynchronize(S()));
        if (p) HIP_CHECK(hipFree(p));
        HIP_CHECK(hipMalloc((void**)&p, bytes));
        cap = bytes;
    }
    return p;
}
static float* nc(const float* p) { return const_cast<float*>(p); }

extern "C" {

const char* dl4ds_last_error(void) { return g_last_error.c_str(); }

int dl4ds_init(int device) {
    API_BEGIN
    {
        std::lock_guard<std::mutex> lk(g_rt_mu);
        DL4DS_REQUIRE(!g_rt.inited || g_rt.device == device, "dl4ds_init: already initialised on another device");
        HIP_CHECK(hipSetDevice(device));
    }
    rt_ensure_init();
    API_END
}
int dl4ds_device_count(int* n) {
    API_BEGIN
    HIP_CHECK(hipGetDeviceCount(n));
    API_END
}
int dl4ds_device_name(char* buf, int buflen) {
    API_BEGIN
    rt_ensure_init();
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, rt().device));
    std::snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    API_END
}
int dl4ds_malloc(void** p, size_t bytes) {
    API_BEGIN
    rt_ensure_init();
    HIP_CHECK(hipMalloc(p, bytes > 0 ? bytes : 4));
    API_END
}
int dl4ds_free(void* p) {
    API_BEGIN
    if (p) { HIP_CHECK(hipStreamSynchronize(S())); HIP_CHECK(hipFree(p)); }
    API_END
}
int dl4ds_memcpy_h2d(void* dst, const void* src, size_t bytes) {
    API_BEGIN
    HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, S()));
    HIP_CHECK(hipStreamSynchronize(S()));
    API_END
}
int dl4ds_memcpy_d2h(void* dst, const void* src, size_t bytes) {
    API_BEGIN
    HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, S()));
    HIP_CHECK(hipStreamSynchronize(S()));
    API_END
}
int dl4ds_host_register(void* p, size_t bytes) {
    API_BEGIN
    DL4DS_REQUIRE(p && bytes, "host_register: empty range");
    HIP_CHECK(hipHostRegister(p, bytes, hipHostRegisterDefault));
    API_END
}
int dl4ds_host_unregister(void* p) {
    API_BEGIN
    HIP_CHECK(hipHostUnregister(p));
    API_END
}
int dl4ds_memcpy_d2d(void* dst, const void* src, size_t bytes) {
    API_BEGIN
    HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, S()));
    API_END
}
int dl4ds_memset(void* p, int value, size_t bytes) {
    API_BEGIN
    HIP_CHECK(hipMemsetAsync(p, value, bytes, S()));
    API_END
}
int dl4ds_sync(void) {
    API_BEGIN
    dist_stream_sync(S(), "dl4ds_sync");          // (plain hipStreamSynchronize unless several ranks take part)
    API_END
}
namespace {
__global__ void raise_device_error_kernel(unsigned* word, unsigned code) {
    __hip_atomic_fetch_or(word, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace
int dl4ds_debug_raise_device_error(int code) {
    API_BEGIN
    hipLaunchKernelGGL(raise_device_error_kernel, dim3(1), dim3(1), 0, S(), device_error_word(), (unsigned)code);
    HIP_CHECK(hipGetLastError());
    API_END
}
int dl4ds_event_timer_start(void) {
    API_BEGIN
    HIP_CHECK(hipEventRecord(rt().ev0, S()));
    API_END
}
int dl4ds_event_timer_stop(float* ms) {
    API_BEGIN
    HIP_CHECK(hipEventRecord(rt().ev1, S()));
    HIP_CHECK(hipEventSynchronize(rt().ev1));
    HIP_CHECK(hipEventElapsedTime(ms, rt().ev0, rt().ev1));
    API_END
}

int dl4ds_profile_enable(int on) {
    API_BEGIN
    HIP_CHECK(hipStreamSynchronize(S()));
    prof().reset();
    prof().on = (on != 0);
    API_END
}
int dl4ds_profile_filter(const char* tag_prefix) {
    API_BEGIN
    prof().filter = tag_prefix ? tag_prefix : "";
    API_END
}
int dl4ds_profile_report(char* buf, size_t buflen) {
    API_BEGIN
    const std::string r = prof().report_json(S());
    DL4DS_REQUIRE(r.size() + 1 <= buflen, "profile report buffer too small");
    std::memcpy(buf, r.c_str(), r.size() + 1);
    API_END
}

// ------------------------------------------------------------------------------------------------ ops
int dl4ds_op_conv2d_fwd(const float* x, const float* w, const float* b, const float* add, float* y, int N, int H, int W,
                        int Cin, int Cout, int KS, int relu, int d2s_r) {
    API_BEGIN
    TView in = make_view(nc(x), N, H, W, Cin);
    TView out = (d2s_r > 1) ? make_view_d2s(y, N, H, W, Cout, d2s_r) : make_view(y, N, H, W, Cout);
    ConvEpilogue ep;
    ep.bias = b;
    if (add) ep.add = make_view(nc(add), N, H, W, Cout);
    ep.relu = relu;
    conv2d_forward(S(), in, w, KS, out, ep);
    API_END
}
int dl4ds_op_conv2d_epilogue(const float* x, const float* w, const float* b, const float* add, const float* mask, float* y, int N,
                             int H, int W, int Cin, int Cout, int KS, int relu, int accumulate) {
    API_BEGIN
    ConvEpilogue ep;
    ep.bias = b;
    if (add) ep.add = make_view(nc(add), N, H, W, Cout);
    if (mask) ep.mask = make_view(nc(mask), N, H, W, Cout);
    ep.relu = relu;
    ep.accumulate = accumulate;
    conv2d_forward(S(), make_view(nc(x), N, H, W, Cin), w, KS, make_view(y, N, H, W, Cout), ep);
    API_END
}
int dl4ds_op_conv2d_dgrad(const float* dz, const float* w, float* dx, int N, int H, int W, int Cin, int Cout, int KS,
                          int d2s_r, int accumulate) {
    API_BEGIN
    const size_t nw = (size_t)KS * KS * Cin * Cout;
    float* wt = scratch(nw * sizeof(float));
    conv2d_dgrad_weights(S(), w, wt, KS, Cin, Cout);
    TView dzv = (d2s_r > 1) ? make_view_d2s(nc(dz), N, H, W, Cout, d2s_r) : make_view(nc(dz), N, H, W, Cout);
    ConvEpilogue ep;
    ep.accumulate = accumulate;
    conv2d_forward(S(), dzv, wt, KS, make_view(dx, N, H, W, Cin), ep);
    API_END
}
int dl4ds_op_conv2d_wgrad(const float* x, const float* dz, float* dw, int N, int H, int W, int Cin, int Cout, int KS,
                          int d2s_r, int accumulate) {
    API_BEGIN
    TView xv = make_view(nc(x), N, H, W, Cin);
    TView dzv = (d2s_r > 1) ? make_view_d2s(nc(dz), N, H, W, Cout, d2s_r) : make_view(nc(dz), N, H, W, Cout);
    const size_t ws = conv2d_wgrad_workspace_bytes(xv, dzv, KS);
    conv2d_wgrad(S(), xv, dzv, KS, dw, accumulate, nullptr, 0, scratch(ws), ws);
    API_END
}
int dl4ds_op_bias_act_bwd(float* dy, const float* y, float* db, int N, int H, int W, int C) {
    API_BEGIN
    TView dyv = make_view(dy, N, H, W, C);
    TView none{nullptr, 0, 0, 0, 0, 0, 0, 0};
    const size_t ws = bias_grad_workspace_bytes(dyv);
    bias_act_backward(S(), dyv, y ? make_view(nc(y), N, H, W, C) : none, y ? dyv : none, db, 0, scratch(ws), ws);
    API_END
}
int dl4ds_op_conv2d_transpose_fwd(const float* x, const float* w, float* y, int N, int H, int W, int Cin, int Cout,
                                  int KS, int stride, int relu) {
    API_BEGIN
    TView in = make_view(nc(x), N, H, W, Cin);
    TView out = make_view(y, N, H * stride, W * stride, Cout);
    const size_t ws = conv2d_transpose_workspace_bytes(in, out, KS, stride);
    conv2d_transpose_forward(S(), in, w, KS, stride, out, relu, scratch(ws), ws);
    API_END
}
int dl4ds_op_conv2d_transpose_dgrad(const float* dz, const float* w, float* dx, int N, int H, int W, int Cin, int Cout,
                                    int KS, int stride, int accumulate) {
    API_BEGIN
    TView dxv = make_view(dx, N, H, W, Cin);
    TView dzv = make_view(nc(dz), N, H * stride, W * stride, Cout);
    const size_t ws = conv2d_transpose_workspace_bytes(dxv, dzv, KS, stride);
    conv2d_transpose_dgrad(S(), dzv, w, KS, stride, dxv, accumulate, scratch(ws), ws);
    API_END
}
int dl4ds_op_conv2d_transpose_wgrad(const float* x, const float* dz, float* dw, int N, int H, int W, int Cin, int Cout,
                                    int KS, int stride, int accumulate) {
    API_BEGIN
    TView xv = make_view(nc(x), N, H, W, Cin);
    TView dzv = make_view(nc(dz), N, H * stride, W * stride, Cout);
    const size_t ws = conv2d_transpose_workspace_bytes(xv, dzv, KS, stride);
    conv2d_transpose_wgrad(S(), xv, dzv, KS, stride, dw, accumulate, scratch(ws), ws);
    API_END
}
int dl4ds_batch_prepare(const float* hr, const float* pred, const float* stat, const int* idx_host, const int* cy_host,
                        const int* cx_host, float* out_lr, float* out_hr, float* out_stat, int H, int W, int C, int P, int S_,
                        int T, int B, int scale, int psy, int psx, int pin, int static_in_lr) {
    API_BEGIN
    DL4DS_REQUIRE(B > 0 && idx_host && cy_host && cx_host, "batch_prepare: index lists missing");
    // the three B-element index lists travel in one small upload (persistent device buffer, grown on demand)
    static int* d_idx = nullptr;
    static int cap = 0;
    static std::vector<int> h_idx;
    if (3 * B > cap) {
        if (d_idx) HIP_CHECK(hipFree(d_idx));
        cap = std::max(3 * B, 3 * 256);
        HIP_CHECK(hipMalloc((void**)&d_idx, (size_t)cap * sizeof(int)));
    }
    // the upload is ordered on the library stream behind the previous batch's kernels, which read the old contents
    h_idx.resize(3 * (size_t)B);
    for (int b = 0; b < B; ++b) { h_idx[b] = idx_host[b]; h_idx[B + b] = cy_host[b]; h_idx[2 * B + b] = cx_host[b]; }
    HIP_CHECK(hipMemcpyAsync(d_idx, h_idx.data(), 3 * (size_t)B * sizeof(int), hipMemcpyHostToDevice, S()));
    batch_prepare(S(), hr, pred, stat, d_idx, d_idx + B, d_idx + 2 * B, out_lr, out_hr, out_stat, H, W, C, P, S_, T, B, scale,
                  psy, psx, pin, static_in_lr);
    API_END
}
int dl4ds_batch_prepare_taps(const float* hr, const float* pred, const float* stat, const int* idx_host, const int* cy_host,
                             const int* cx_host, float* out_lr, float* out_hr, float* out_stat, float* scratch_dev, int H, int W,
                             int C, int P, int S_, int T, int B, int scale, int psy, int psx, int pin, int static_in_lr,
                             const dl4ds_tap_axis* dn_patch, const dl4ds_tap_axis* dn_field, const dl4ds_tap_axis* up_field) {
    API_BEGIN
    DL4DS_REQUIRE(B > 0 && idx_host && cy_host && cx_host, "batch_prepare_taps: index lists missing");
    static int* d_idx = nullptr;
    static int cap = 0;
    static std::vector<int> h_idx;
    if (3 * B > cap) {
        if (d_idx) HIP_CHECK(hipFree(d_idx));
        cap = std::max(3 * B, 3 * 256);
        HIP_CHECK(hipMalloc((void**)&d_idx, (size_t)cap * sizeof(int)));
    }
    h_idx.resize(3 * (size_t)B);
    for (int b = 0; b < B; ++b) { h_idx[b] = idx_host[b]; h_idx[B + b] = cy_host[b]; h_idx[2 * B + b] = cx_host[b]; }
    HIP_CHECK(hipMemcpyAsync(d_idx, h_idx.data(), 3 * (size_t)B * sizeof(int), hipMemcpyHostToDevice, S()));
    auto axes = [](const dl4ds_tap_axis* t, TapAxis* o) -> const TapAxis* {
        if (!t) return nullptr;
        for (int i = 0; i < 2; ++i) { o[i].idx = t[i].idx; o[i].wt = t[i].wt; o[i].k = t[i].k; }
        return o;
    };
    TapAxis a0[2], a1[2], a2[2];
    batch_prepare_taps(S(), hr, pred, stat, d_idx, d_idx + B, d_idx + 2 * B, out_lr, out_hr, out_stat, scratch_dev, H, W, C, P,
                       S_, T, B, scale, psy, psx, pin, static_in_lr, axes(dn_patch, a0), axes(dn_field, a1), axes(up_field, a2));
    API_END
}
int dl4ds_batch_gather(const dl4ds_gather_group* groups, int n_groups, const int* idx_host, const int* cy_host, const int* cx_host,
                       float* out_dev, int out_h, int out_w, int T, int B) {
    API_BEGIN
    DL4DS_REQUIRE(groups && n_groups >= 1 && n_groups <= 3 && B > 0, "batch_gather: groups / batch size");
    DL4DS_REQUIRE((cy_host == nullptr) == (cx_host == nullptr), "batch_gather: crop corner lists");
    static int* d_idx = nullptr;
    static int cap = 0;
    static std::vector<int> h_idx;
    if (3 * B > cap) {
        if (d_idx) HIP_CHECK(hipFree(d_idx));
        cap = std::max(3 * B, 3 * 256);
        HIP_CHECK(hipMalloc((void**)&d_idx, (size_t)cap * sizeof(int)));
    }
    h_idx.assign(3 * (size_t)B, 0);
    for (int b = 0; b < B; ++b) {
        if (idx_host) h_idx[b] = idx_host[b];
        if (cy_host) { h_idx[B + b] = cy_host[b]; h_idx[2 * B + b] = cx_host[b]; }
    }
    HIP_CHECK(hipMemcpyAsync(d_idx, h_idx.data(), 3 * (size_t)B * sizeof(int), hipMemcpyHostToDevice, S()));
    GatherGroup g[3];
    for (int i = 0; i < n_groups; ++i) {
        const dl4ds_gather_group& q = groups[i];
        DL4DS_REQUIRE(q.src_dev && q.channels > 0 && q.src_h > 0 && q.src_w > 0 && q.row_div >= 0, "batch_gather: group");
        // the kernel does not clamp: every crop has to stay inside its source (raw groups: the corner + the output extent; a patch that
        // was resized: the corner inside the field -- its tap indices are the caller's table, relative to the corner)
        for (int b = 0; b < B && (q.raw || q.origin_from_crop); ++b) {
            const int cy = cy_host ? cy_host[b] : 0, cx = cx_host ? cx_host[b] : 0;
            const int sy = q.row_div ? cy / q.row_div : cy, sx = q.row_div ? cx / q.row_div : cx;
            DL4DS_REQUIRE(cy >= 0 && cx >= 0 && sy < q.src_h && sx < q.src_w, "batch_gather: crop corner outside the source");
            if (q.raw) DL4DS_REQUIRE(sy + out_h <= q.src_h && sx + out_w <= q.src_w, "batch_gather: crop leaves the source");
        }
        g[i].src = q.src_dev; g[i].channels = q.channels; g[i].frames = q.frames; g[i].src_h = q.src_h; g[i].src_w = q.src_w;
        g[i].raw = q.raw; g[i].origin_from_crop = q.origin_from_crop; g[i].row_div = q.row_div;
        for (int k = 0; k < 2; ++k) { g[i].taps[k].idx = q.taps[k].idx; g[i].taps[k].wt = q.taps[k].wt; g[i].taps[k].k = q.taps[k].k; }
    }
    batch_gather(S(), g, n_groups, idx_host ? d_idx : nullptr, cy_host ? d_idx + B : nullptr, cy_host ? d_idx + 2 * B : nullptr, out_dev,
                 out_h, out_w, T, B);
    API_END
}
int dl4ds_op_depth_to_space(const float* x, float* y, int N, int H, int W, int C, int r) {
    API_BEGIN
    depth_to_space(S(), x, y, N, H, W, C, r);
    API_END
}
int dl4ds_op_space_to_depth(const float* y, float* x, int N, int H, int W, int C, int r) {
    API_BEGIN
    space_to_depth(S(), y, x, N, H, W, C, r);
    API_END
}
int dl4ds_op_maxpool2_fwd(const float* x, float* y, int N, int H, int W, int C) {
    API_BEGIN
    maxpool2_forward(S(), make_view(nc(x), N, H, W, C), make_view(y, N, H / 2, W / 2, C));
    API_END
}
int dl4ds_op_maxpool2_bwd(const float* x, const float* y, const float* dy, float* dx, int N, int H, int W, int C) {
    API_BEGIN
    int acc = 0;
    if ((H | W) & 1) { fill(S(), dx, (size_t)N * H * W * C, 0.f); acc = 1; }
    maxpool2_backward(S(), make_view(nc(x), N, H, W, C), make_view(nc(y), N, H / 2, W / 2, C),
                      make_view(nc(dy), N, H / 2, W / 2, C), make_view(dx, N, H, W, C), acc);
    API_END
}
int dl4ds_op_dwconv_fwd(const float* x, const float* k, const float* bias, float* y, int N, int H, int W, int C, int KS) {
    API_BEGIN
    dwconv_forward(S(), x, k, bias, y, N, H, W, C, KS, 0, 0);
    API_END
}
int dl4ds_op_dwconv_bwd(const float* x, const float* k, const float* dy, float* dx, float* dk, float* db, int N, int H, int W, int C,
                        int KS, int accumulate) {
    API_BEGIN
    if (dx) dwconv_forward(S(), dy, k, nullptr, dx, N, H, W, C, KS, 1, accumulate);
    if (dk) {
        const size_t ws = dwconv_wgrad_workspace_bytes(C, KS);
        dwconv_wgrad(S(), x, dy, dk, db, accumulate, N, H, W, C, KS, scratch(ws), ws);
    }
    API_END
}
int dl4ds_op_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, size_t npix, int C, float eps,
                           int relu) {
    API_BEGIN
    layernorm_forward(S(), x, gamma, beta, y, npix, C, eps, relu);
    API_END
}
int dl4ds_op_layernorm_bwd(const float* x, const float* y, const float* dy, const float* gamma, float* dx, float* dgamma,
                           float* dbeta, size_t npix, int C, float eps, int relu, int accumulate) {
    API_BEGIN
    const size_t ws = norm_workspace_bytes(C);
    layernorm_backward(S(), x, y, dy, gamma, dx, accumulate, dgamma, dbeta, accumulate, npix, C, eps, relu, scratch(ws), ws);
    API_END
}
int dl4ds_op_batchnorm_fwd(const float* x, const float* gamma, const float* beta, float* moving_mean, float* moving_var,
                           float* y, float* saved, size_t npix, int C, float eps, float momentum, int training, int relu) {
    API_BEGIN
    const size_t ws = norm_workspace_bytes(C);
    batchnorm_forward(S(), x, gamma, beta, moving_mean, moving_var, y, saved, npix, C, eps, momentum, training, relu,
                      scratch(ws), ws);
    API_END
}
int dl4ds_op_batchnorm_bwd(const float* x, const float* y, const float* dy, const float* gamma, const float* saved, float* dx,
                           float* dgamma, float* dbeta, size_t npix, int C, int relu, int accumulate) {
    API_BEGIN
    const size_t ws = norm_workspace_bytes(C);
    batchnorm_backward(S(), x, y, dy, gamma, saved, dx, accumulate, dgamma, dbeta, accumulate, npix, C, relu, scratch(ws), ws);
    API_END
}
int dl4ds_op_resize_bilinear_fwd(const float* x, float* y, int N, int H, int W, int C, int Ho, int Wo) {
    API_BEGIN
    resize_bilinear_forward(S(), make_view(nc(x), N, H, W, C), make_view(y, N, Ho, Wo, C));
    API_END
}
int dl4ds_op_resize_bilinear_bwd(const float* dy, float* dx, int N, int H, int W, int C, int Ho, int Wo) {
    API_BEGIN
    resize_bilinear_backward(S(), make_view(nc(dy), N, Ho, Wo, C), make_view(dx, N, H, W, C), 0);
    API_END
}
int dl4ds_op_localconv_fwd(const float* x, const float* w, const float* b, float* y, int N, int H, int W, int C, int F) {
    API_BEGIN
    localconv_forward(S(), make_view(nc(x), N, H, W, C), w, b, make_view(y, N, H, W, F));
    API_END
}
int dl4ds_op_localconv_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int N,
                           int H, int W, int C, int F) {
    API_BEGIN
    TView dxv{nullptr, 0, 0, 0, 0, 0, 0, 0};
    if (dx) dxv = make_view(dx, N, H, W, C);
    localconv_backward(S(), make_view(nc(x), N, H, W, C), w, make_view(nc(dy), N, H, W, F), dxv, 0, dw, db, 0);
    API_END
}
int dl4ds_op_chatt_fwd(const float* x, float* y, int G, int R, int P, int C, int Cr, const float* w1, const float* b1,
                       const float* w2, const float* b2, float* saved) {
    API_BEGIN
    AttShape sh{G, R, P, C, Cr};
    const size_t inst = (size_t)G * P;
    float* mean = saved;
    float* scale = mean + inst * C;
    float* hidden = scale + inst * C;
    chatt_forward(S(), x, y, sh, w1, b1, w2, b2, mean, hidden, scale, scratch(chatt_workspace_bytes(sh)));
    API_END
}
int dl4ds_op_chatt_bwd(const float* x, const float* dy, float* dx, int G, int R, int P, int C, int Cr, const float* w1,
                       const float* w2, const float* saved, float* dw1, float* db1, float* dw2, float* db2) {
    API_BEGIN
    AttShape sh{G, R, P, C, Cr};
    const size_t inst = (size_t)G * P;
    const float* mean = saved;
    const float* scale = mean + inst * C;
    const float* hidden = scale + inst * C;
    chatt_backward(S(), x, dy, dx, 0, sh, w1, w2, mean, hidden, scale, dw1, db1, dw2, db2, 0,
                   scratch(chatt_workspace_bytes(sh)));
    API_END
}
int dl4ds_op_loss(int kind, const float* yt, const float* yp, float* dpred, int N, int H, int W, int C, float* loss_dev) {
    API_BEGIN
    const size_t ws = loss_workspace_bytes(kind, N, H, W, C);
    loss_forward_backward(S(), kind, yt, yp, dpred, N, H, W, C, 1.f, loss_dev, 0, scratch(ws), ws);
    API_END
}
int dl4ds_metrics(const float* yt, const float* yp, int N, int H, int W, int C, float* pair_out_dev, float* grid_out_dev,
                  float* range_out_dev) {
    API_BEGIN
    const size_t ws = metrics_workspace_bytes(N, H, W, C);
    image_metrics(S(), yt, yp, N, H, W, C, pair_out_dev, grid_out_dev, range_out_dev, scratch(ws), ws);
    API_END
}
int dl4ds_op_bce(const float* p, float label, int n, float* loss_dev, float* dp) {
    API_BEGIN
    bce_forward_backward(S(), p, label, n, 1.f, loss_dev, dp, 0);
    API_END
}
int dl4ds_op_adam(float* w, const float* g, float* m, float* v, size_t n, int t, float lr, float beta1, float beta2,
                  float eps, float grad_scale) {
    API_BEGIN
    const float lr_t = (float)((double)lr * std::sqrt(1.0 - std::pow((double)beta2, (double)t)) /
                               (1.0 - std::pow((double)beta1, (double)t)));
    adam_update(S(), w, g, m, v, n, lr_t, beta1, beta2, eps, grad_scale, device_error_word_if_any());       // (skips the update once the sticky word is set)
    API_END
}

// ------------------------------------------------------------------------------------------------ graph
int dl4ds_graph_create(dl4ds_graph** g) {
    API_BEGIN
    *g = new dl4ds_graph();
    (*g)->g.stream = S();
    // Weight-gradient kernels on a second stream (concurrent with the dgrad chain) paid off with the LDS-staged conv
    // kernels (+2-3 %); with the streamed-filter / row-walking kernels two co-resident MFMA kernels only take matrix-core
    // and L1 bandwidth from each other (-2 % measured), so the single-stream order is the default now.
    if (getenv("DL4DS_AUX_STREAM")) (*g)->g.aux_stream = rt().aux_stream;
    API_END
}
int dl4ds_graph_destroy(dl4ds_graph* g) {
    API_BEGIN
    if (g) { HIP_CHECK(hipStreamSynchronize(S())); delete g; }
    API_END
}
int dl4ds_graph_input(dl4ds_graph* g, int H, int W, int C, int nmul, int* tid) {
    API_BEGIN
    *tid = g->g.add_tensor(H, W, C, nmul, false, true);
    API_END
}
int dl4ds_graph_input_requires_grad(dl4ds_graph* g, int tid) {
    API_BEGIN
    DL4DS_REQUIRE(!g->g.finalized, "graph already finalized");
    GTensor& t = g->g.tensors.at(tid);
    DL4DS_REQUIRE(t.is_input, "not an input tensor");
    t.requires_grad = true;
    t.dep_grad_input = true;
    API_END
}
int dl4ds_graph_param(dl4ds_graph* g, size_t n, int* pid) {
    API_BEGIN
    *pid = g->g.add_param(n);
    API_END
}
int dl4ds_graph_conv2d(dl4ds_graph* g, int in, int w, int b, int add, int KS, int Cout, int relu, int d2s_r, int* out) {
    API_BEGIN
    *out = g_conv2d(g->g, in, w, b, add, KS, Cout, relu, d2s_r);
    API_END
}
int dl4ds_graph_conv2d_transpose(dl4ds_graph* g, int in, int w, int KS, int stride, int Cout, int relu, int* out) {
    API_BEGIN
    *out = g_conv2d_transpose(g->g, in, w, KS, stride, Cout, relu);
    API_END
}
int dl4ds_graph_chatt(dl4ds_graph* g, int in, int w1, int b1, int w2, int b2, int Cr, int T5, int* out) {
    API_BEGIN
    *out = g_chatt(g->g, in, w1, b1, w2, b2, Cr, T5);
    API_END
}
int dl4ds_graph_concat(dl4ds_graph* g, const int* ins, int n, int* out) {
    API_BEGIN
    *out = g_concat(g->g, ins, n);
    API_END
}
int dl4ds_graph_add(dl4ds_graph* g, int a, int b, int relu, int* out) {
    API_BEGIN
    *out = g_add(g->g, a, b, relu);
    API_END
}
int dl4ds_graph_act(dl4ds_graph* g, int in, int kind, int* out) {
    API_BEGIN
    *out = g_act(g->g, in, kind);
    API_END
}
int dl4ds_graph_maxpool2(dl4ds_graph* g, int in, int* out) {
    API_BEGIN
    *out = g_maxpool2(g->g, in);
    API_END
}
int dl4ds_graph_resize(dl4ds_graph* g, int in, int Ho, int Wo, int* out) {
    API_BEGIN
    *out = g_resize(g->g, in, Ho, Wo);
    API_END
}
int dl4ds_graph_resize_nearest(dl4ds_graph* g, int in, int Ho, int Wo, int* out) {
    API_BEGIN
    *out = g_resize(g->g, in, Ho, Wo, 1);
    API_END
}
int dl4ds_graph_resize_bicubic(dl4ds_graph* g, int in, int Ho, int Wo, int* out) {
    API_BEGIN
    *out = g_resize(g->g, in, Ho, Wo, 2);
    API_END
}
int dl4ds_graph_resize_method(dl4ds_graph* g, int in, int Ho, int Wo, int method, int* out) {
    API_BEGIN
    *out = g_resize(g->g, in, Ho, Wo, method);
    API_END
}
int dl4ds_graph_localconv(dl4ds_graph* g, int in, int w, int b, int F, int* out) {
    API_BEGIN
    *out = g_localconv(g->g, in, w, b, F);
    API_END
}
int dl4ds_rec_tail_supported(int CX, int CS, int CO, int* yes) {
    API_BEGIN
    *yes = rec_tail_supported(CX, CS, CO) ? 1 : 0;
    API_END
}
int dl4ds_graph_rec_tail(dl4ds_graph* g, int x, int s, int wt, int bt, int wl, int bl, int w, int b, int T, int CO, int* out) {
    API_BEGIN
    *out = g_rec_tail(g->g, x, s, wt, bt, wl, bl, w, b, T, CO);
    API_END
}
int dl4ds_graph_repeat_time(dl4ds_graph* g, int in, int T, int* out) {
    API_BEGIN
    *out = g_repeat_time(g->g, in, T);
    API_END
}
int dl4ds_graph_convlstm(dl4ds_graph* g, int in, int wk, int wr, int b, int KS, int F, int T, int relu, int* out) {
    API_BEGIN
    *out = g_convlstm(g->g, in, wk, wr, b, KS, F, T, relu);
    API_END
}
int dl4ds_graph_gap(dl4ds_graph* g, int in, int* out) {
    API_BEGIN
    *out = g_gap(g->g, in, 0);
    API_END
}
int dl4ds_graph_dwconv(dl4ds_graph* g, int in, int w, int b, int KS, int* out) {
    API_BEGIN
    *out = g_dwconv(g->g, in, w, b, KS);
    API_END
}
int dl4ds_graph_gap3d(dl4ds_graph* g, int in, int* out) {
    API_BEGIN
    *out = g_gap(g->g, in, 1);
    API_END
}
int dl4ds_graph_conv2d_folded(dl4ds_graph* g, int in, int w1, int b1, int w2, int b2, int KS, int Cmid, int Cout, int relu, int d2s,
                              int* out) {
    API_BEGIN
    *out = g_conv2d_folded(g->g, in, w1, b1, w2, b2, KS, Cmid, Cout, relu, d2s);
    API_END
}
int dl4ds_graph_conv2d_folded_aux(dl4ds_graph* g, int in, int aux, int w1, int b1, int w2, int b2, int KS, int Cmid, int Cout,
                                  int relu, int d2s, int* out) {
    API_BEGIN
    *out = g_conv2d_folded(g->g, in, w1, b1, w2, b2, KS, Cmid, Cout, relu, d2s, aux);
    API_END
}
int dl4ds_graph_pad(dl4ds_graph* g, int in, int Ho, int Wo, int* out) {
    API_BEGIN
    *out = g_pad(g->g, in, Ho, Wo);
    API_END
}
int dl4ds_graph_slice(dl4ds_graph* g, int in, int oy, int ox, int step, int Ho, int Wo, int* out) {
    API_BEGIN
    *out = g_slice(g->g, in, oy, ox, step, Ho, Wo);
    API_END
}
int dl4ds_graph_dense(dl4ds_graph* g, int in, int w, int b, int F, int act, int* out) {
    API_BEGIN
    *out = g_dense(g->g, in, w, b, F, act);
    API_END
}
int dl4ds_graph_dropout(dl4ds_graph* g, int in, float rate, int* out) {
    API_BEGIN
    *out = g_dropout(g->g, in, rate);
    API_END
}
int dl4ds_graph_dropout_variant(dl4ds_graph* g, int in, float rate, int variant, int mc, int spatial_dim, int* out) {
    API_BEGIN
    *out = g_dropout(g->g, in, rate, variant, mc, spatial_dim);
    API_END
}
int dl4ds_graph_dropout_count(dl4ds_graph* g, int* n) {
    API_BEGIN
    *n = (int)g->g.dropout_ops.size();
    API_END
}
static GOp* dropout_op(dl4ds_graph* g, int index, int B) {
    DL4DS_REQUIRE(index >= 0 && index < (int)g->g.dropout_ops.size(), "dropout op index out of range");
    DL4DS_REQUIRE(g->g.finalized && B > 0 && B <= g->g.maxB, "dropout mask: graph not prepared for this batch size");
    return g->g.dropout_ops[index];
}
int dl4ds_graph_dropout_mask_size(dl4ds_graph* g, int index, int B, size_t* n) {
    API_BEGIN
    DL4DS_REQUIRE(index >= 0 && index < (int)g->g.dropout_ops.size(), "dropout op index out of range");
    *n = g->g.dropout_ops[index]->mask_floats(g->g, B);
    API_END
}
int dl4ds_graph_dropout_get_mask(dl4ds_graph* g, int index, int B, float* dst_host) {
    API_BEGIN
    GOp* op = dropout_op(g, index, B);
    HIP_CHECK(hipMemcpyAsync(dst_host, op->saved, op->mask_floats(g->g, B) * sizeof(float), hipMemcpyDeviceToHost, g->g.stream));
    HIP_CHECK(hipStreamSynchronize(g->g.stream));
    API_END
}
int dl4ds_graph_dropout_set_mask(dl4ds_graph* g, int index, int B, const float* src_host) {
    API_BEGIN
    GOp* op = dropout_op(g, index, B);
    op->set_mask(g->g, src_host, op->mask_floats(g->g, B));
    HIP_CHECK(hipStreamSynchronize(g->g.stream));
    API_END
}
int dl4ds_graph_norm(dl4ds_graph* g, int in, int gamma, int beta, int mov_mean, int mov_var, int batch, float eps, int relu,
                     int* out) {
    API_BEGIN
    *out = g_norm(g->g, in, gamma, beta, mov_mean, mov_var, batch, eps, relu);
    API_END
}
int dl4ds_graph_output(dl4ds_graph* g, int tid) {
    API_BEGIN
    DL4DS_REQUIRE(tid >= 0 && tid < (int)g->g.tensors.size(), "bad tensor id");
    g->g.outputs.push_back(tid);
    API_END
}
int dl4ds_graph_finalize(dl4ds_graph* g) {
    API_BEGIN
    g->g.finalize();
    API_END
}
int dl4ds_graph_tensor_shape(dl4ds_graph* g, int tid, int shape4[4]) {
    API_BEGIN
    const GTensor& t = g->g.tensors.at(tid);
    shape4[0] = t.nmul; shape4[1] = t.H; shape4[2] = t.W; shape4[3] = t.C;
    API_END
}
int dl4ds_graph_param_count(dl4ds_graph* g, size_t* n_arena, int* n_params) {
    API_BEGIN
    *n_arena = g->g.n_params;
    *n_params = (int)g->g.params.size();
    API_END
}
int dl4ds_graph_param_info(dl4ds_graph* g, int pid, size_t* offset, size_t* n) {
    API_BEGIN
    *offset = g->g.params.at(pid).offset;
    *n = g->g.params.at(pid).n;
    API_END
}
int dl4ds_graph_set_param(dl4ds_graph* g, int pid, const float* src) {
    API_BEGIN
    DL4DS_REQUIRE(g->g.finalized, "graph not finalized");
    const GParam& p = g->g.params.at(pid);
    HIP_CHECK(hipMemcpyAsync(g->g.W + p.offset, src, p.n * sizeof(float), hipMemcpyHostToDevice, S()));
    HIP_CHECK(hipStreamSynchronize(S()));
    API_END
}
int dl4ds_graph_get_param(dl4ds_graph* g, int pid, float* dst) {
    API_BEGIN
    const GParam& p = g->g.params.at(pid);
    HIP_CHECK(hipMemcpyAsync(dst, g->g.W + p.offset, p.n * sizeof(float), hipMemcpyDeviceToHost, S()));
    HIP_CHECK(hipStreamSynchronize(S()));
    API_END
}
int dl4ds_graph_get_grad(dl4ds_graph* g, int pid, float* dst) {
    API_BEGIN
    const GParam& p = g->g.params.at(pid);
    HIP_CHECK(hipMemcpyAsync(dst, g->g.G + p.offset, p.n * sizeof(float), hipMemcpyDeviceToHost, S()));
    HIP_CHECK(hipStreamSynchronize(S()));
    API_END
}
int dl4ds_graph_arena_ptrs(dl4ds_graph* g, float** w, float** gr) {
    API_BEGIN
    *w = g->g.W;
    *gr = g->g.G;
    API_END
}
int dl4ds_graph_forward(dl4ds_graph* g, const float* const* inputs, int n_inputs, int B, int training, int is_host,
                        float* out) {
    API_BEGIN
    Graph& G = g->g;
    DL4DS_REQUIRE(!G.outputs.empty(), "graph has no output");
    graph_load_inputs(G, inputs, n_inputs, B, is_host != 0);
    G.forward(B, training != 0);
    if (out) {
        const GTensor& o = G.tensors[G.outputs[0]];
        HIP_CHECK(hipMemcpyAsync(out, o.data, o.per_sample() * B * sizeof(float),
                                 is_host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, S()));
        if (is_host) HIP_CHECK(hipStreamSynchronize(S()));
    }
    API_END
}
int dl4ds_graph_fusion_report(dl4ds_graph* g, int B, char* json_buf, size_t buflen) {
    API_BEGIN
    g->g.prepare(B);                       // fusion decisions need the allocated buffers (alignment of the real views)
    std::string out = "[";
    for (auto& op : g->g.ops) {
        const std::string d = op->describe_fusion(g->g);
        if (d.empty()) continue;
        if (out.size() > 1) out += ",";
        out += d;
    }
    out += "]";
    DL4DS_REQUIRE(out.size() + 1 <= buflen, "fusion_report: buffer too small");
    std::memcpy(json_buf, out.c_str(), out.size() + 1);
    API_END
}
int dl4ds_graph_bucket_plan(dl4ds_graph* g, char* buf, size_t buflen) {
    API_BEGIN
    DL4DS_REQUIRE(g && g->g.finalized, "graph not finalized");
    // launch order = the order in which the backward pass completes the buckets (largest ready_op first)
    std::vector<int> order(g->g.buckets.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return g->g.buckets[a].ready_op > g->g.buckets[b].ready_op; });
    std::string r = "[";
    for (size_t i = 0; i < order.size(); ++i) {
        const auto& b = g->g.buckets[order[i]];
        if (i) r += ",";
        r += "{\"bytes\":" + std::to_string(b.n * sizeof(float)) + ",\"offset\":" + std::to_string(b.off * sizeof(float)) +
             ",\"params\":" + std::to_string(b.p_hi - b.p_lo) + ",\"final_after_backward_of_op\":" + std::to_string(b.ready_op) +
             ",\"of_ops\":" + std::to_string(g->g.ops.size()) + "}";
    }
    r += "]";
    DL4DS_REQUIRE(r.size() + 1 <= buflen, "bucket plan buffer too small");
    std::memcpy(buf, r.c_str(), r.size() + 1);
    API_END
}
int dl4ds_graph_tensor_ptr(dl4ds_graph* g, int tid, int grad, float** p) {
    API_BEGIN
    const GTensor& t = g->g.tensors.at(tid);
    *p = grad ? t.grad : t.data;
    API_END
}

// ------------------------------------------------------------------------------------------------ training
int dl4ds_trainer_create(dl4ds_graph* g, int loss_kind, float lr0, float lr1, double lr_boundary, float beta1,
                         float beta2, float eps, dl4ds_trainer** tr) {
    API_BEGIN
    AdamCfg c;
    c.lr0 = lr0; c.lr1 = lr1; c.boundary = lr_boundary; c.beta1 = beta1; c.beta2 = beta2; c.eps = eps;
    dl4ds_trainer* h = new dl4ds_trainer();
    h->t = trainer_create(&g->g, loss_kind, c);
    *tr = h;
    API_END
}
int dl4ds_trainer_destroy(dl4ds_trainer* tr) {
    API_BEGIN
    if (tr) {
        HIP_CHECK(hipStreamSynchronize(S()));
        if (tr->t) delete tr->t;
        if (tr->c) cgan_destroy(tr->c);
        delete tr;
    }
    API_END
}
int dl4ds_trainer_step(dl4ds_trainer* tr, const float* const* inputs, int n_inputs, const float* y_true, int B,
                       int is_host, float* loss_host) {
    API_BEGIN
    DL4DS_REQUIRE(tr && tr->t, "not a supervised trainer");
    trainer_step(*tr->t, inputs, n_inputs, y_true, B, is_host != 0, loss_host);
    API_END
}
int dl4ds_trainer_loss_and_grads(dl4ds_trainer* tr, const float* const* inputs, int n_inputs, const float* y_true,
                                 int B, int is_host, float* loss_host) {
    API_BEGIN
    DL4DS_REQUIRE(tr && tr->t, "not a supervised trainer");
    trainer_loss_and_grads(*tr->t, inputs, n_inputs, y_true, B, is_host != 0, false);    // this rank's gradients
    if (loss_host) {
        HIP_CHECK(hipMemcpyAsync(loss_host, tr->t->d_loss, sizeof(float), hipMemcpyDeviceToHost, S()));
        HIP_CHECK(hipStreamSynchronize(S()));
    }
    API_END
}
int dl4ds_trainer_evaluate(dl4ds_trainer* tr, const float* const* inputs, int n_inputs, const float* y_true, int B,
                           int is_host, float* loss_host) {
    API_BEGIN
    DL4DS_REQUIRE(tr && tr->t, "not a supervised trainer");
    DL4DS_REQUIRE(loss_host, "evaluate: loss_host is required");
    trainer_evaluate(*tr->t, inputs, n_inputs, y_true, B, is_host != 0);
    HIP_CHECK(hipMemcpyAsync(loss_host, tr->t->d_loss, sizeof(float), hipMemcpyDeviceToHost, S()));
    dist_stream_sync(S(), "dl4ds_trainer_evaluate");
    API_END
}
int dl4ds_trainer_get_state(dl4ds_trainer* tr, float* m_host, float* v_host, long* step) {
    API_BEGIN
    DL4DS_REQUIRE(tr && tr->t, "not a supervised trainer");
    Trainer& t = *tr->t;
    const size_t bytes = t.g->n_params * sizeof(float);
    if (m_host) HIP_CHECK(hipMemcpyAsync(m_host, t.m, bytes, hipMemcpyDeviceToHost, S()));
    if (v_host) HIP_CHECK(hipMemcpyAsync(v_host, t.v, bytes, hipMemcpyDeviceToHost, S()));
    HIP_CHECK(hipStreamSynchronize(S()));
    if (step) *step = t.step;
    API_END
}
int dl4ds_trainer_set_state(dl4ds_trainer* tr, const float* m_host, const float* v_host, long step) {
    API_BEGIN
    DL4DS_REQUIRE(tr && tr->t, "not a supervised trainer");
    DL4DS_REQUIRE(m_host && v_host && step >= 0, "set_state: missing arrays / negative step");
    Trainer& t = *tr->t;
    const size_t bytes = t.g->n_params * sizeof(float);
    HIP_CHECK(hipMemcpyAsync(t.m, m_host, bytes, hipMemcpyHostToDevice, S()));
    HIP_CHECK(hipMemcpyAsync(t.v, v_host, bytes, hipMemcpyHostToDevice, S()));
    HIP_CHECK(hipStreamSynchronize(S()));
    t.step = step;
    API_END
}
static Trainer& cgan_side(dl4ds_trainer* tr, int which) {
    DL4DS_REQUIRE(tr && tr->c, "not a CGAN trainer");
    DL4DS_REQUIRE(which == 0 || which == 1, "which: 0 generator, 1 discriminator");
    return which == 0 ? *cgan_gen_trainer(tr->c) : *cgan_disc_trainer(tr->c);
}
int dl4ds_cgan_get_state(dl4ds_trainer* tr, int which, float* m_host, float* v_host, long* step) {
    API_BEGIN
    Trainer& t = cgan_side(tr, which);
    const size_t bytes = t.g->n_params * sizeof(float);
    if (m_host) HIP_CHECK(hipMemcpyAsync(m_host, t.m, bytes, hipMemcpyDeviceToHost, S()));
    if (v_host) HIP_CHECK(hipMemcpyAsync(v_host, t.v, bytes, hipMemcpyDeviceToHost, S()));
    HIP_CHECK(hipStreamSynchronize(S()));
    if (step) *step = t.step;
    API_END
}
int dl4ds_cgan_set_state(dl4ds_trainer* tr, int which, const float* m_host, const float* v_host, long step) {
    API_BEGIN
    Trainer& t = cgan_side(tr, which);
    DL4DS_REQUIRE(m_host && v_host && step >= 0, "set_state: missing arrays / negative step");
    const size_t bytes = t.g->n_params * sizeof(float);
    HIP_CHECK(hipMemcpyAsync(t.m, m_host, bytes, hipMemcpyHostToDevice, S()));
    HIP_CHECK(hipMemcpyAsync(t.v, v_host, bytes, hipMemcpyHostToDevice, S()));
    HIP_CHECK(hipStreamSynchronize(S()));
    t.step = step;
    API_END
}
int dl4ds_trainer_last_loss(dl4ds_trainer* tr, float* loss_host) {
    API_BEGIN
    DL4DS_REQUIRE(tr && tr->t, "not a supervised trainer");
    HIP_CHECK(hipMemcpyAsync(loss_host, tr->t->d_loss, sizeof(float), hipMemcpyDeviceToHost, S()));
    dist_stream_sync(S(), "dl4ds_trainer_last_loss");
    API_END
}

int dl4ds_cgan_create(dl4ds_graph* gen, dl4ds_graph* disc, int px_loss_kind, float lr, float beta1, float lam,
                      dl4ds_trainer** tr) {
    API_BEGIN
    dl4ds_trainer* h = new dl4ds_trainer();
    h->c = cgan_create(&gen->g, &disc->g, px_loss_kind, lr, beta1, lam);
    *tr = h;
    API_END
}
int dl4ds_cgan_set_learning_rates(dl4ds_trainer* tr, float gen_lr, float disc_lr) {
    API_BEGIN
    DL4DS_REQUIRE(tr && tr->c, "not a CGAN trainer");
    DL4DS_REQUIRE(gen_lr > 0.f && disc_lr > 0.f, "learning rates must be positive");
    cgan_set_learning_rates(tr->c, gen_lr, disc_lr);
    API_END
}
int dl4ds_cgan_step(dl4ds_trainer* tr, const float* const* gen_inputs, int n_gen_inputs, const float* hr, int B,
                    int is_host, const float* dropout_keep_host, int apply_update, float* losses_host) {
    API_BEGIN
    DL4DS_REQUIRE(tr && tr->c, "not a CGAN trainer");
    cgan_step(*tr->c, gen_inputs, n_gen_inputs, hr, B, is_host != 0, dropout_keep_host, apply_update != 0, losses_host);
    API_END
}
int dl4ds_cgan_get_disc_grad(dl4ds_trainer* tr, int pid, float* dst) {
    API_BEGIN
    DL4DS_REQUIRE(tr && tr->c, "not a CGAN trainer");
    Graph* g = cgan_disc_trainer(tr->c)->g;
    const GParam& p = g->params.at(pid);
    HIP_CHECK(hipMemcpyAsync(dst, g->G + p.offset, p.n * sizeof(float), hipMemcpyDeviceToHost, S()));
    HIP_CHECK(hipStreamSynchronize(S()));
    API_END
}

// ------------------------------------------------------------------------------------------------ dist
int dl4ds_dist_unique_id(char id128[128]) {
    API_BEGIN
    dist_unique_id(id128);
    API_END
}
int dl4ds_dist_init(int rank, int world, const char id128[128]) {
    API_BEGIN
    dist_init(rank, world, id128);
    API_END
}
int dl4ds_dist_world(int* rank, int* world) {
    API_BEGIN
    dist_world(*rank, *world);
    API_END
}
int dl4ds_dist_broadcast_trainer(dl4ds_trainer* tr, int root) {
    API_BEGIN
    DL4DS_REQUIRE(tr, "broadcast_trainer: null trainer");
    dist_require_ready("dl4ds_dist_broadcast_trainer");
    std::vector<Trainer*> ts;
    if (tr->t) ts.push_back(tr->t);
    if (tr->c) { ts.push_back(cgan_gen_trainer(tr->c)); ts.push_back(cgan_disc_trainer(tr->c)); }
    for (Trainer* t : ts) {
        dist_broadcast(t->g->W, t->g->n_params, root, S());
        dist_broadcast(t->m, t->g->n_params, root, S());
        dist_broadcast(t->v, t->g->n_params, root, S());
    }
    HIP_CHECK(hipStreamSynchronize(S()));
    for (Trainer* t : ts) dist_broadcast_i64(&t->step, root);     // optimizer.iterations (a resumed rank 0)
    API_END
}
int dl4ds_dist_expected_world(int* world) {
    API_BEGIN
    *world = dist_expected_world();
    API_END
}
int dl4ds_dist_comm_info(int* nranks, int* rank, int* device) {
    API_BEGIN
    dist_comm_info(*nranks, *rank, *device);
    API_END
}
int dl4ds_dist_allreduce_host(float* values_host, int n, int op) {
    API_BEGIN
    rt_ensure_init();
    dist_allreduce_host(values_host, n, op);
    API_END
}
int dl4ds_dist_barrier(void) {
    API_BEGIN
    rt_ensure_init();
    dist_barrier();
    API_END
}
int dl4ds_dist_allreduce_sum(float* buf, size_t n) {
    API_BEGIN
    dist_allreduce_grads(buf, n, S());
    API_END
}
int dl4ds_dist_finalize(void) {
    API_BEGIN
    dist_finalize();
    API_END
}

}  // extern "C"
