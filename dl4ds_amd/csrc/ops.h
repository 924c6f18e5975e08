// Internal C++ launch API of the gfx950 kernels (one stream argument everywhere; no hidden syncs).
#pragma once
#include "common.h"

// ----------------------------------------------------------------------------- conv (conv.hip)
// y = epilogue(conv_same_stride1(x, w)) ; w is HWIO flattened [KS*KS][Cin][Cout]
struct ConvEpilogue {
    const float* bias = nullptr;   // [Cout] or null
    TView add{nullptr, 0, 0, 0, 0, 0, 0, 0};    // optional residual added before the activation
    TView mask{nullptr, 0, 0, 0, 0, 0, 0, 0};   // optional: result zeroed where mask <= 0 (ReLU backward)
    int relu = 0;
    int accumulate = 0;            // out += result (gradient accumulation)
    // optional [tiles][8] per-tile channel sums of the stored values (tiles of one image contiguous, see
    // conv2d_narrow_pair_tiles_per_image): the pooling of a ChannelAttention2D that consumes this output.  Only the
    // narrow pair kernel emits it; callers check conv2d_narrow_pair_ok first, every other path rejects it.
    float* pool = nullptr;
};
void conv2d_forward(hipStream_t s, const TView& in, const float* w, int KS, const TView& out,
                    const ConvEpilogue& ep);
// wt[(KS*KS-1-tap)][co][ci] = w[tap][ci][co] : conv2d_forward(dz, wt) == dgrad
void conv2d_dgrad_weights(hipStream_t s, const float* w, float* wt, int KS, int Cin, int Cout);
// several of them in one launch: jobs_dev = nj x DgradWeightsJob on the device, blocks = sum of their block counts
struct DgradWeightsJob { const float* src; float* dst; int KK, Cin, Cout, block0; };
int dgrad_weights_job_blocks(int KK, int Cin, int Cout);
void conv2d_dgrad_weights_batched(hipStream_t s, const DgradWeightsJob* jobs_dev, int nj, int blocks);
// dw[tap][ci][co] (+)= sum_{n,y,x} x[n,y+ky-p,x+kx-p,ci] * dz[n,y,x,co] ; db[co] (+)= sum dz[n,y,x,co] (db may be null)
size_t conv2d_wgrad_workspace_bytes(const TView& x, const TView& dz, int KS);
void conv2d_wgrad(hipStream_t s, const TView& x, const TView& dz, int KS, float* dw, int accumulate, float* db,
                  int accumulate_db, float* workspace, size_t workspace_bytes);
// Stencil (VALU) path for 3x3 layers with pad2(Cin)*pad2(Cout) <= 8 (conv_direct.hip); conv2d_forward / conv2d_wgrad
// route to it themselves.  forward returns false when the layer is not eligible; wgrad_slabs returns 0.
bool conv2d_direct_forward(hipStream_t s, const TView& in, const float* w, int KS, const TView& out,
                           const ConvEpilogue& ep);
// 3x3, Cin <= 16, Cout <= 16: register-resident filter, persistent MFMA kernel (conv_narrow.hip)
bool conv2d_narrow_forward(hipStream_t s, const TView& in, const float* w, int KS, const TView& out,
                           const ConvEpilogue& ep);
// 1x1 layers whose channel counts are not multiples of four (conv_point.hip): contiguous staging + MFMA + contiguous store
bool conv2d_point_forward(hipStream_t s, const TView& in, const float* w, int KS, const TView& out, const ConvEpilogue& ep);
bool conv2d_narrow_pair_ok(const TView& in, const TView& out, int KS, const ConvEpilogue& ep);
int conv2d_narrow_pair_tiles_per_image(int H, int W);
bool conv2d_direct_eligible(const TView& in, const TView& out, int KS);
// weight gradient of a stencil convolution behind a ChannelAttention2D scale + the attention's d(loss)/d(scale), both from
// per-image raw weight gradients (conv_direct.hip); false: not eligible, nothing was launched
bool conv2d_direct_wgrad_attention(hipStream_t s, const TView& x_raw, const TView& dz, int KS, const float* scale, const float* w,
                                   float* dw, int accumulate, float* db, int accumulate_db, float* ds, float* workspace,
                                   size_t workspace_bytes);
// 3x3 wgrad with Cin <= 8, Cout <= 16 (two taps per MFMA tile); same slab protocol as the direct path
int conv2d_narrow_wgrad_slabs(const TView& x, const TView& dz, int KS);
int conv2d_narrow_wgrad(hipStream_t s, const TView& x, const TView& dz, int KS, float* partial, int max_slabs);
// 3x3, Cin >= 16: filter streamed from L2 into the MFMA operands, no barriers in the K loop (conv_stream.hip)
// small grids (H W <= max_hw) with many channels as a GEMM over flattened pixels (conv_gemm.hip); false = not eligible
bool conv2d_gemm_forward(hipStream_t s, const TView& in, const float* w, int KS, const TView& out, const ConvEpilogue& ep, int max_hw);
// Winograd F(2x2, 3x3) form of the MFMA-bound 3x3 layers (conv_wino.hip); false = not eligible
bool conv2d_wino_forward(hipStream_t s, const TView& in, const float* w, const TView& out, const ConvEpilogue& ep);
// the 40 / 48-channel 3x3 layers with their fp32 products as six bf16 MFMA terms (conv_split.hip); false = not eligible / DL4DS_NO_SPLIT
bool conv2d_split_forward(hipStream_t s, const TView& in, const float* w, const TView& out, const ConvEpilogue& ep);
// The transformed filters of a graph's Winograd layers, one batched launch per pass instead of one per layer (conv_wino.hip):
// a pass of a graph holds a WinoPassGuard (kind 0 = forward, 1 = backward); _invalidate marks every registered filter inside
// [lo, hi) stale, _refresh transforms the stale ones of that kind in one launch on s, _release frees them (graph destruction).
struct WinoPassGuard { int prev_kind; explicit WinoPassGuard(int kind); ~WinoPassGuard(); WinoPassGuard(const WinoPassGuard&) = delete; };
void wino_filters_invalidate(const float* lo, const float* hi);
void wino_filters_refresh(hipStream_t s, const float* lo, const float* hi, int kind);
void wino_filters_release(const float* lo, const float* hi);
bool wino_pass_active(int& kind);                              // inside a graph pass? (kind: 0 forward, 1 backward)
// conv_split.hip: the same three for the six-term kernel's filter fragments (called by the wino_filters_* functions)
void split_filters_invalidate(const float* lo, const float* hi);
void split_filters_refresh(hipStream_t s, const float* lo, const float* hi, int kind);
void split_filters_release(const float* lo, const float* hi);
// ... and of their weight gradient (conv_wino_wgrad.hip): dw [3][3][Cin][Cout], db [Cout] or null
bool conv2d_wino_wgrad(hipStream_t s, const TView& x, const TView& dy, float* dw, int accumulate, float* db, int accumulate_db);
bool conv2d_stream_forward(hipStream_t s, const TView& in, const float* w, int KS, const TView& out,
                           const ConvEpilogue& ep);
int conv2d_direct_wgrad_slabs(const TView& x, const TView& dz, int KS);
// writes at most `slabs` partial slabs (the workspace bound); returns how many it wrote
int conv2d_direct_wgrad(hipStream_t s, const TView& x, const TView& dz, int KS, float* partial, int slabs);
// Conv2DTranspose(k, stride s, 'same', no bias), kernel HWOI [k*k][Cout][Cin] (blocks.py:508-516)
void conv2d_transpose_forward(hipStream_t s, const TView& in, const float* w, int KS, int stride,
                              const TView& out, int relu, float* workspace, size_t workspace_bytes);
// `relu_mask` (may be null): the layer's INPUT activation; its gradient is zeroed where that activation is <= 0 (the ReLU
// backward of the producing layer folded into this store)
void conv2d_transpose_dgrad(hipStream_t s, const TView& dz, const float* w, int KS, int stride,
                            const TView& dx, int accumulate, float* workspace, size_t workspace_bytes,
                            const TView* relu_mask = nullptr);
void conv2d_transpose_wgrad(hipStream_t s, const TView& x, const TView& dz, int KS, int stride,
                            float* dw, int accumulate, float* workspace, size_t workspace_bytes);
size_t conv2d_transpose_workspace_bytes(const TView& in, const TView& out, int KS, int stride);

// ------------------------------------------------------------------- elementwise (elementwise.hip)
// dz = dy * (y > 0 ? 1 : 0) (if y.p) written to dz (may alias dy); db[c] (+)= sum dz[...,c] (if db)
size_t bias_grad_workspace_bytes(const TView& dy);
void bias_act_backward(hipStream_t s, const TView& dy, const TView& y, const TView& dz, float* db,
                       int accumulate_db, float* workspace, size_t workspace_bytes);
// dst (+)= alpha * src  (same logical shape; either side may be a strided / d2s view)
void view_axpy(hipStream_t s, const TView& src, const TView& dst, float alpha, int accumulate);
// dst (+)= src * [mask > 0] (mask.p == nullptr: plain copy): Concatenate backward of a ReLU output whose mask the consumers apply
void view_axpy_masked(hipStream_t s, const TView& src, const TView& mask, const TView& dst, int accumulate);
// Concatenate backward in one pass over the wide gradient (elementwise.hip): slice k = channels [off, off + C) -> dense dst (+ mask)
struct ConcatSlice { float* dst; const float* mask; int off, C, accumulate; };
void concat_split(hipStream_t s, const float* src, int ld, size_t npx, const ConcatSlice* slices, int n);
// the mirror, Concatenate forward in one pass: slices[k].dst is read as the dense input k (mask / accumulate unused)
void concat_join(hipStream_t s, float* dst, int ld, size_t npx, const ConcatSlice* slices, int n);
// dst (+)= dy * [y > 0]  (flat, contiguous)
void masked_axpy(hipStream_t s, const float* dy, const float* y, float* dst, size_t n, int accumulate);
// ... for both operands of an Add at once (false: not applicable, call masked_axpy twice)
bool masked_axpy_pair(hipStream_t s, const float* dy, const float* ya, float* da, int acc_a, const float* yb, float* db, int acc_b, size_t n);
// out = act(a + b)
void add_act(hipStream_t s, const float* a, const float* b, float* out, size_t n, int relu);
// in-place / out-of-place activation forward y = f(x) and backward dx (+)= dy * f'(x)
enum ActKind { ACT_NONE = 0, ACT_RELU = 1, ACT_SIGMOID = 2, ACT_TANH = 3, ACT_ELU = 4,
               ACT_LEAKY_RELU = 5, ACT_SELU = 6, ACT_GELU = 7 };
void act_forward(hipStream_t s, const float* x, float* y, size_t n, int kind);
void act_backward(hipStream_t s, const float* x, const float* dy, float* dx, size_t n, int kind,
                  int accumulate);
void fill(hipStream_t s, float* p, size_t n, float v);
// DepthwiseConv2D(7, 'same') of ConvNextBlock (dwconv.hip).  flip != 0: mirrored taps = the input gradient.
void dwconv_forward(hipStream_t s, const float* x, const float* k, const float* bias, float* y, int N, int H, int W, int C,
                    int KS, int flip, int accumulate);
size_t dwconv_wgrad_workspace_bytes(int C, int KS);
void dwconv_wgrad(hipStream_t s, const float* x, const float* dy, float* dk, float* db, int accumulate, int N, int H, int W, int C,
                  int KS, float* ws, size_t ws_bytes);
// post-hoc test metrics of dl4ds/metrics.py:166-262 (dssim.hip): pair_out [N][4] = (mae, mse, pearson over the grid, ssim),
// grid_out [3][H*W*C] = (rmse, mean bias, pearson over the pairs), range_out [2] = joint (min, max)
size_t metrics_workspace_bytes(int N, int H, int W, int C);
void image_metrics(hipStream_t s, const float* y_true, const float* y_pred, int N, int H, int W, int C, float* pair_out_dev,
                   float* grid_out_dev, float* range_out_dev, float* workspace, size_t workspace_bytes);
// LayerNormalization / BatchNormalization over the channel axis of [npix][C] (norm.hip), optional fused ReLU
size_t norm_workspace_bytes(int C);
void layernorm_forward(hipStream_t s, const float* x, const float* gamma, const float* beta, float* y, size_t npix, int C,
                       float eps, int relu);
void layernorm_backward(hipStream_t s, const float* x, const float* y, const float* dy, const float* gamma, float* dx, int acc_dx,
                        float* dgamma, float* dbeta, int acc_dw, size_t npix, int C, float eps, int relu, float* ws,
                        size_t ws_bytes);
void batchnorm_forward(hipStream_t s, const float* x, const float* gamma, const float* beta, float* mov_mean, float* mov_var,
                       float* y, float* saved, size_t npix, int C, float eps, float momentum, int training, int relu, float* ws,
                       size_t ws_bytes);
void batchnorm_backward(hipStream_t s, const float* x, const float* y, const float* dy, const float* gamma, const float* saved,
                        float* dx, int acc_dx, float* dgamma, float* dbeta, int acc_dw, size_t npix, int C, int relu, float* ws,
                        size_t ws_bytes);
// depth_to_space standalone (used by tests and unfused fallbacks)
void depth_to_space(hipStream_t s, const float* x, float* y, int N, int H, int W, int C, int r);
void space_to_depth(hipStream_t s, const float* y, float* x, int N, int H, int W, int C, int r);
// MaxPooling2D(2,2) valid
void maxpool2_forward(hipStream_t s, const TView& x, const TView& y);
// relu_mask != 0: x is a ReLU output whose backward rides on this store (gradient zeroed where x <= 0)
void maxpool2_backward(hipStream_t s, const TView& x, const TView& y, const TView& dy, const TView& dx,
                       int accumulate, int relu_mask = 0);
// bilinear resize (half-pixel centres)
void resize_bilinear_forward(hipStream_t s, const TView& x, const TView& y);
void resize_nearest_forward(hipStream_t s, const TView& x, const TView& y);
void resize_nearest_backward(hipStream_t s, const TView& dy, const TView& dx, int accumulate);
void resize_bilinear_backward(hipStream_t s, const TView& dy, const TView& dx, int accumulate);
// LocallyConnected2D 1x1: y[n,h,w,f] = b[h,w,f] + sum_c x[n,h,w,c] W[h,w,c,f]
void localconv_forward(hipStream_t s, const TView& x, const float* w, const float* b, const TView& y);
void localconv_backward(hipStream_t s, const TView& x, const float* w, const TView& dy, const TView& dx,
                        int accumulate_dx, float* dw, float* db, int accumulate_dw);

// --------------------------------------------------------- channel attention (attention.hip)
// x viewed as [G][R][Q]: mean over R, MLP over the last C channels of Q=(P*C), scale.
struct AttShape { int G, R, P, C, Cr; };
// pool_partial (optional): [G][pool_tiles][8] per-tile channel sums emitted by the producing convolution (ConvEpilogue::pool)
// -> no pooling pass over x.  y == nullptr: the scale is not applied here (the consumer reads x through TView::sc = scale).
void chatt_forward(hipStream_t s, const float* x, float* y, const AttShape& sh, const float* w1,
                   const float* b1, const float* w2, const float* b2, float* mean, float* hidden,
                   float* scale, float* workspace, const float* pool_partial = nullptr, int pool_tiles = 0);
// dx == nullptr: dX = dY * scale + dmean is not materialised; dmean is left in dmean_out ([G*P*C], caller-owned) for the
// producer's lazy dY view (TView::sc = scale, TView::sh = dmean)
void chatt_backward(hipStream_t s, const float* x, const float* dy, float* dx, int accumulate_dx,
                    const AttShape& sh, const float* w1, const float* w2, const float* mean,
                    const float* hidden, const float* scale, float* dw1, float* db1, float* dw2,
                    float* db2, int accumulate_dw, float* workspace, float* dmean_out = nullptr,
                    const float* ds_given = nullptr);     // ds_given ([G*P*C]): sum_r(dy * x) already known -> no pass over dy, x
size_t chatt_workspace_bytes(const AttShape& sh);

// ----------------------------------------------------------------------- losses (losses.hip)
enum LossKind { LOSS_MAE = 0, LOSS_MSE = 1, LOSS_DSSIM = 2, LOSS_DSSIM_MAE = 3, LOSS_DSSIM_MSE = 4,
                LOSS_DSSIM_MAE_MSE = 5, LOSS_MSDSSIM = 6, LOSS_MSDSSIM_MAE = 7, LOSS_MSDSSIM_MAE_MSE = 8 };
// loss_out[0] = scale * loss ; dpred (+)= scale * dloss/dpred.  y_true,y_pred: (N,H,W,C) contiguous.
size_t loss_workspace_bytes(int kind, int N, int H, int W, int C);
void loss_forward_backward(hipStream_t s, int kind, const float* y_true, const float* y_pred,
                           float* dpred, int N, int H, int W, int C, float scale, float* loss_out,
                           int accumulate, float* workspace, size_t workspace_bytes);
// BCE on probabilities p[n] vs constant label; loss_out[0] (+)= scale*mean ; dp = scale*dL/dp
void bce_forward_backward(hipStream_t s, const float* p, float label, int n, float scale,
                          float* loss_out, float* dp, int accumulate_loss);

// -------------------------------------------------------------------------- optimiser (adam.hip)
// Keras Adam on a flat arena. lr_t = lr*sqrt(1-b2^t)/(1-b1^t) computed by the caller.
void adam_update(hipStream_t s, float* w, const float* g, float* m, float* v, size_t n, float lr_t,
                 float beta1, float beta2, float eps, float grad_scale, const unsigned* err_word = nullptr);   // err_word: see runtime.h

// ------------------------------------------------------------- batch preparation (batchprep.hip), "next" row f1
// Gathers one training batch from a device-resident dataset: block-mean coarsening (cv2 INTER_AREA at an integer
// ratio), pixel replication back to the HR grid for 'pin', crops, predictor / static-variable channel stacking.
// idx / cy / cx are DEVICE int arrays of length B (first frame, crop corner in HR pixels).
void batch_prepare(hipStream_t s, const float* hr, const float* pred, const float* stat, const int* idx, const int* cy,
                   const int* cx, float* out_lr, float* out_hr, float* out_stat, int H, int W, int C, int P, int S, int T,
                   int B, int scale, int psy, int psx, int pin, int static_in_lr);
// One axis of a separable cv2.resize: k (source index, weight) taps per output row / column, device arrays [n_out][k].
struct TapAxis { const int* idx; const float* wt; int k; };
// The same for any interpolation (see batchprep.hip): each table argument is {y axis, x axis}.
void batch_prepare_taps(hipStream_t s, const float* hr, const float* pred, const float* stat, const int* idx, const int* cy,
                        const int* cx, float* out_lr, float* out_hr, float* out_stat, float* scratch, int H, int W, int C,
                        int P, int S, int T, int B, int scale, int psy, int psx, int pin, int static_in_lr,
                        const TapAxis* dn_patch, const TapAxis* dn_field, const TapAxis* up_field);
// one gather pass over up to three channel groups (raw crops or separable tap tables): the primitive behind both of the above
struct GatherGroup { const float* src; int channels, frames, src_h, src_w, raw, origin_from_crop, row_div; TapAxis taps[2]; };
void batch_gather(hipStream_t s, const GatherGroup* groups, int n_groups, const int* idx, const int* cy, const int* cx, float* out,
                  int out_h, int out_w, int T, int B);
void repeat_time_forward(hipStream_t s, const float* in, float* out, int B, int T, size_t ps);
void repeat_time_forward_view(hipStream_t s, const float* in, const TView& out, int B, int T);     // out: (B*T, H, W, C) view, any pixel pitch
void repeat_time_backward(hipStream_t s, const float* dout, float* din, int B, int T, size_t ps, int accumulate);
void repeat_time_backward_view(hipStream_t s, const TView& dout, float* din, int B, int T, int accumulate);   // dout: slice view
void resize_table_forward(hipStream_t s, const TView& x, const TView& y, const int* iy, const float* wy, const int* ix, const float* wx,
                          int ky, int kx);   // [out][k] taps per axis
void resize_table_backward(hipStream_t s, const TView& dy, const TView& dx, const int* py, const int* oy, const float* vy,
                           const int* px, const int* ox, const float* vx, int accumulate, int max_taps_x = 0);
