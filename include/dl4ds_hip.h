/* dl4ds_hip.h -- C ABI of libdl4ds_hip.so: the MI355X (gfx950) native replacement for the
 * TensorFlow/Keras + Horovod arithmetic behind dl4ds's conv-SR train step.
 *
 * The reference (carlos-gg/dl4ds 1.8.0) has NO plugin / FFI interface: its hot path sits behind plain
 * Python (`dl4ds.models.*` builders returning tf.keras.Model; `SupervisedTrainer.run`, `CGANTrainer.run`,
 * `Predictor.run`).  The entry points below are what a ctypes binding for that path needs; each names the
 * reference code it replaces (paths relative to the reference repo).  The Python mirror that binds them is
 * dl4ds_amd/ (same builder / trainer signatures as the reference).
 *
 * Conventions: every function returns 0 on success, <0 on error (message: dl4ds_last_error()); no C++
 * exception crosses the boundary; all tensors are fp32 NHWC ("channels_last"), conv kernels HWIO,
 * transposed-conv kernels HWOI, dense kernels [in][out]; pointers named *_dev are device (HBM) pointers,
 * *_host are host pointers; sizes are element counts unless called bytes.  One library-wide HIP stream;
 * calls are asynchronous unless documented otherwise.  ONE HOST THREAD per process drives the library (one process per GPU): the profiler and a few run-time statics are unsynchronised.
 */
#ifndef DL4DS_HIP_H
#define DL4DS_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dl4ds_graph dl4ds_graph;
typedef struct dl4ds_trainer dl4ds_trainer;

/* ---------------------------------------------------------------- runtime / memory (plumbing)
 * replaces: tf.config device selection, dl4ds/utils.py:174-203; training/base.py:97-122 */
const char* dl4ds_last_error(void);
int dl4ds_init(int device);                       /* hipSetDevice + create the library stream */
int dl4ds_device_count(int* n);
int dl4ds_device_name(char* buf, int buflen);
int dl4ds_malloc(void** p_dev, size_t bytes);
int dl4ds_free(void* p_dev);
int dl4ds_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes);   /* synchronous */
int dl4ds_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes);   /* synchronous */
int dl4ds_memcpy_d2d(void* dst_dev, const void* src_dev, size_t bytes);    /* async on the stream */
/* Page-lock a host buffer the caller owns for the duration of a run of host <-> device copies (hipHostRegister / hipHostUnregister):
 * copies to / from it then go at the link's rate instead of through the runtime's staging of pageable memory.  Used by
 * Model.predict for the array it returns (inference.py:238-249 hands numpy arrays in and out).  register returns non-zero when the
 * range cannot be pinned (the caller simply goes on with pageable copies). */
int dl4ds_host_register(void* p_host, size_t bytes);
int dl4ds_host_unregister(void* p_host);
int dl4ds_memset(void* p_dev, int value, size_t bytes);
int dl4ds_sync(void);                             /* hipStreamSynchronize(library stream); fails if a kernel raised the
                                                   * sticky device-side error word since the last wait (e.g. the persistent
                                                   * ConvLSTM kernel gave up waiting for a neighbouring tile: that step is invalid) */
/* diagnostic: a one-thread kernel on the library stream raises the device-side error word with `code` exactly as a kernel
 * would; the next host-side wait (dl4ds_sync, a loss read-back ...) must fail once and clear it.  No reference counterpart
 * (TensorFlow reports device-side failures through its own status plumbing). */
int dl4ds_debug_raise_device_error(int code);
int dl4ds_event_timer_start(void);                /* hipEventRecord on the library stream */
int dl4ds_event_timer_stop(float* ms);            /* record + synchronize + elapsed ms */
/* per-launch HIP-event timing on the library stream (off by default).  report: JSON
 * {"<kernel tag>": {"n": launches, "ms": total, "flops": algorithmic, "bytes": algorithmic}, ...};
 * synchronises and clears the log. */
int dl4ds_profile_enable(int on);
/* restrict the timing to launches whose tag starts with `tag_prefix` (NULL / "" = all): lets a benchmark time its
 * dominant kernel inside the measured region without paying two events for every other launch */
int dl4ds_profile_filter(const char* tag_prefix);
int dl4ds_profile_report(char* json_buf, size_t buflen);

/* ---------------------------------------------------------------- single-op entry points
 * (unit-testable against the oracle; contiguous NHWC device tensors) */

/* y = [relu](conv_same_s1(x,w) + b + add), optionally stored through depth_to_space(d2s_r).
 * replaces tf.keras.layers.Conv2D (blocks.py:49-61,208,249-259,299,414-416,479; sp_postups.py:134,156)
 * fused with Add (blocks.py:228), Activation (blocks.py:75) and tf.nn.depth_to_space (blocks.py:427). */
int dl4ds_op_conv2d_fwd(const float* x_dev, const float* w_dev, const float* b_dev, const float* add_dev,
                        float* y_dev, int N, int H, int W, int Cin, int Cout, int KS, int relu, int d2s_r);
/* The same convolution with EVERY operand of the fused epilogue, in the order the train step applies them:
 * y = [y_old +] mask_gt0( [relu]( conv_same_s1(x,w) + b + add ), mask ) -- residual Add (blocks.py:228), Activation
 * (blocks.py:75), the ReLU backward of the layer below riding on a dgrad store (mask = that layer's activation) and gradient
 * accumulation.  The narrow kernels compile one form per operand combination (csrc/conv_narrow.hip); this entry point lets a
 * test reach all of them.  b, add, mask may be NULL; accumulate != 0 adds the result to what y holds. */
int dl4ds_op_conv2d_epilogue(const float* x_dev, const float* w_dev, const float* b_dev, const float* add_dev,
                             const float* mask_dev, float* y_dev, int N, int H, int W, int Cin, int Cout, int KS, int relu,
                             int accumulate);
/* dx (+)= dgrad(dz, w); dz may be given in depth_to_space(d2s_r) layout (gradient of a fused-d2s conv) */
int dl4ds_op_conv2d_dgrad(const float* dz_dev, const float* w_dev, float* dx_dev, int N, int H, int W,
                          int Cin, int Cout, int KS, int d2s_r, int accumulate);
/* dw (+)= wgrad(x, dz) */
int dl4ds_op_conv2d_wgrad(const float* x_dev, const float* dz_dev, float* dw_dev, int N, int H, int W,
                          int Cin, int Cout, int KS, int d2s_r, int accumulate);
/* dz = dy * [y>0] (if y_dev) in place over dy; db = sum_pixels dz (if db_dev) */
int dl4ds_op_bias_act_bwd(float* dy_dev, const float* y_dev, float* db_dev, int N, int H, int W, int C);
/* Conv2DTranspose(k=KS, stride, 'same', use_bias=False) -- blocks.py:508-516; kernel HWOI */
int dl4ds_op_conv2d_transpose_fwd(const float* x_dev, const float* w_dev, float* y_dev, int N, int H, int W,
                                  int Cin, int Cout, int KS, int stride, int relu);
int dl4ds_op_conv2d_transpose_dgrad(const float* dz_dev, const float* w_dev, float* dx_dev, int N, int H,
                                    int W, int Cin, int Cout, int KS, int stride, int accumulate);
int dl4ds_op_conv2d_transpose_wgrad(const float* x_dev, const float* dz_dev, float* dw_dev, int N, int H,
                                    int W, int Cin, int Cout, int KS, int stride, int accumulate);
/* tf.nn.depth_to_space / its adjoint -- blocks.py:427 */
int dl4ds_op_depth_to_space(const float* x_dev, float* y_dev, int N, int H, int W, int C, int r);
int dl4ds_op_space_to_depth(const float* y_dev, float* x_dev, int N, int H, int W, int C, int r);
/* MaxPooling2D((2,2)) -- blocks.py:613 */
int dl4ds_op_maxpool2_fwd(const float* x_dev, float* y_dev, int N, int H, int W, int C);
int dl4ds_op_maxpool2_bwd(const float* x_dev, const float* y_dev, const float* dy_dev, float* dx_dev, int N,
                          int H, int W, int C);
/* DepthwiseConv2D(kernel_size=7, padding='same', depth_multiplier=1) -- ConvNextBlock.dwconv, blocks.py:143-144.
 * k: (7,7,C) taps (Keras (7,7,C,1)), bias (C) or NULL.  bwd: dx / dk (+ db) may be NULL; accumulate adds into them. */
int dl4ds_op_dwconv_fwd(const float* x_dev, const float* k_dev, const float* bias_dev, float* y_dev, int N, int H, int W, int C,
                        int KS);
int dl4ds_op_dwconv_bwd(const float* x_dev, const float* k_dev, const float* dy_dev, float* dx_dev, float* dk_dev, float* db_dev,
                        int N, int H, int W, int C, int KS, int accumulate);
/* LayerNormalization(axis=-1) / BatchNormalization(axis=-1) over [npix][C] -- blocks.py:63-71,151-159,293-296.
 * relu: fuse the activation that follows.  bwd: y is only read when relu != 0; dx / dgamma / dbeta may be NULL;
 * accumulate != 0 adds into dx, dgamma, dbeta.  batchnorm: `saved` (2*C floats: batch mean, 1/std) is written by a
 * training-mode forward and read by the backward; moving statistics are updated in place when training. */
int dl4ds_op_layernorm_fwd(const float* x_dev, const float* gamma_dev, const float* beta_dev, float* y_dev, size_t npix, int C,
                           float eps, int relu);
int dl4ds_op_layernorm_bwd(const float* x_dev, const float* y_dev, const float* dy_dev, const float* gamma_dev, float* dx_dev,
                           float* dgamma_dev, float* dbeta_dev, size_t npix, int C, float eps, int relu, int accumulate);
int dl4ds_op_batchnorm_fwd(const float* x_dev, const float* gamma_dev, const float* beta_dev, float* moving_mean_dev,
                           float* moving_var_dev, float* y_dev, float* saved_dev, size_t npix, int C, float eps,
                           float momentum, int training, int relu);
int dl4ds_op_batchnorm_bwd(const float* x_dev, const float* y_dev, const float* dy_dev, const float* gamma_dev,
                           const float* saved_dev, float* dx_dev, float* dgamma_dev, float* dbeta_dev, size_t npix, int C,
                           int relu, int accumulate);
/* Resizing(..., 'bilinear') -- blocks.py:489; discriminator.py:62-63 */
int dl4ds_op_resize_bilinear_fwd(const float* x_dev, float* y_dev, int N, int H, int W, int C, int Ho, int Wo);
int dl4ds_op_resize_bilinear_bwd(const float* dy_dev, float* dx_dev, int N, int H, int W, int C, int Ho, int Wo);
/* LocallyConnected2D(F,(1,1),implementation=3) -- blocks.py:322-328 */
int dl4ds_op_localconv_fwd(const float* x_dev, const float* w_dev, const float* b_dev, float* y_dev, int N,
                           int H, int W, int C, int F);
int dl4ds_op_localconv_bwd(const float* x_dev, const float* w_dev, const float* dy_dev, float* dx_dev,
                           float* dw_dev, float* db_dev, int N, int H, int W, int C, int F);
/* ChannelAttention2D -- blocks.py:585-593.  x is [G][R][P*C] (4-D: G=B,R=H*W,P=1; 5-D: G=B,R=T*H,P=W).
 * saved_dev: G*P*(2C+Cr) floats written by fwd and consumed by bwd. */
int dl4ds_op_chatt_fwd(const float* x_dev, float* y_dev, int G, int R, int P, int C, int Cr, const float* w1_dev,
                       const float* b1_dev, const float* w2_dev, const float* b2_dev, float* saved_dev);
int dl4ds_op_chatt_bwd(const float* x_dev, const float* dy_dev, float* dx_dev, int G, int R, int P, int C, int Cr,
                       const float* w1_dev, const float* w2_dev, const float* saved_dev, float* dw1_dev,
                       float* db1_dev, float* dw2_dev, float* db2_dev);
/* dl4ds/losses.py:5-149.  kind: 0 mae 1 mse 2 dssim 3 dssim_mae 4 dssim_mse 5 dssim_mae_mse
 * 6 msdssim 7 msdssim_mae 8 msdssim_mae_mse (tf.image.ssim_multiscale, four scales: grids of at least 81x81).
 * loss_dev[0] = loss ; dpred_dev = dloss/dpred (may be NULL). */
int dl4ds_op_loss(int kind, const float* y_true_dev, const float* y_pred_dev, float* dpred_dev, int N, int H,
                  int W, int C, float* loss_dev);
/* compute_metrics (metrics.py:166-262) without the plots, on device-resident test arrays (N,H,W,C):
 *   pair_out_dev  [N][4]       per test pair: MAE, MSE, Pearson correlation over the grid, SSIM (tf.image.ssim with the
 *                              joint dynamic range; NaN when the grid is smaller than the 11x11 window).
 *                              PSNR = 10 log10(range^2 / MSE) follows on the host (tf.image.psnr, metrics.py:168-169)
 *   grid_out_dev  [3][H*W*C]   per grid point over the pairs: RMSE (metrics.py:188), mean bias (:219), Pearson (:247)
 *   range_out_dev [2]          joint (min, max) of both arrays (metrics.py:166) */
int dl4ds_metrics(const float* y_true_dev, const float* y_pred_dev, int N, int H, int W, int C, float* pair_out_dev,
                  float* grid_out_dev, float* range_out_dev);
/* Keras BinaryCrossentropy(from_logits=False) vs a constant label -- cgan.py:546-549,567-571 */
int dl4ds_op_bce(const float* p_dev, float label, int n, float* loss_dev, float* dp_dev);
/* tf.keras.optimizers.Adam step t (1-based) -- supervised.py:353; cgan.py:277-278 */
int dl4ds_op_adam(float* w_dev, const float* g_dev, float* m_dev, float* v_dev, size_t n, int t, float lr,
                  float beta1, float beta2, float eps, float grad_scale);

/* ---------------------------------------------------------------- model graph
 * replaces the tf.keras functional graphs assembled by dl4ds/models/{sp_postups,sp_preups,spt_postups,
 * spt_preups,discriminator}.py.  The Python builders add tensors / parameters / ops in call order. */
int dl4ds_graph_create(dl4ds_graph** g);
int dl4ds_graph_destroy(dl4ds_graph* g);
/* nmul: batch multiplier of the tensor (1, or time_window for (B,T,H,W,C) tensors) */
int dl4ds_graph_input(dl4ds_graph* g, int H, int W, int C, int nmul, int* tensor_id);
/* allocate a gradient buffer for an input (the discriminator's HR input: the generator's adversarial
 * gradient flows through it -- cgan.py:600-613) */
int dl4ds_graph_input_requires_grad(dl4ds_graph* g, int tensor_id);
int dl4ds_graph_param(dl4ds_graph* g, size_t n, int* param_id);
int dl4ds_graph_conv2d(dl4ds_graph* g, int in, int w, int b, int add, int KS, int Cout, int relu, int d2s_r, int* out);
int dl4ds_graph_conv2d_transpose(dl4ds_graph* g, int in, int w, int KS, int stride, int Cout, int relu, int* out);
int dl4ds_graph_chatt(dl4ds_graph* g, int in, int w1, int b1, int w2, int b2, int Cr, int time_window_5d, int* out);
int dl4ds_graph_concat(dl4ds_graph* g, const int* ins, int n, int* out);
int dl4ds_graph_add(dl4ds_graph* g, int a, int b, int relu, int* out);
int dl4ds_graph_act(dl4ds_graph* g, int in, int kind, int* out);   /* 1 relu 2 sigmoid 3 tanh 4 elu 5 leaky 6 selu 7 gelu */
int dl4ds_graph_maxpool2(dl4ds_graph* g, int in, int* out);
int dl4ds_graph_resize(dl4ds_graph* g, int in, int Ho, int Wo, int* out);
/* Resizing(Ho, Wo, interpolation='nearest') (half-pixel centres) -- ResizeConvolutionBlock(interpolation='nearest'), blocks.py:473-489 */
int dl4ds_graph_resize_nearest(dl4ds_graph* g, int in, int Ho, int Wo, int* out);
/* Resizing(interpolation='bicubic') = tf.image.resize(method='bicubic'): ResizeBicubic with half-pixel centres (Keys cubic,
 * A = -0.5, 1024-step weight table, out-of-image taps dropped and the rest renormalised) -- blocks.py:473-489 */
int dl4ds_graph_resize_bicubic(dl4ds_graph* g, int in, int Ho, int Wo, int* out);
/* Resizing(interpolation=...) by number: 0 bilinear, 1 nearest, 2 bicubic, and tf.image.resize's ScaleAndTranslate family
 * (antialias=False: scale = out / in, kernel scale 1, spans clamped into the image and normalised): 3 lanczos3, 4 lanczos5,
 * 5 gaussian (radius 1.5, sigma 0.5), 6 mitchellcubic -- blocks.py:473-489.  ('area' has no gradient in TensorFlow, so a
 * ResizeConvolutionBlock built with it cannot be trained by the reference either; it is rejected.) */
int dl4ds_graph_resize_method(dl4ds_graph* g, int in, int Ho, int Wo, int method, int* out);
int dl4ds_graph_localconv(dl4ds_graph* g, int in, int w, int b, int F, int* out);
int dl4ds_graph_repeat_time(dl4ds_graph* g, int in, int T, int* out);
/* The recurrent nets' tail as ONE op -- spt_postups.py:133-151 / spt_preups.py:114-132, blocks.py:301-333:
 *   y = TransitionLast(Concatenate([x24, LocalizedConvBlock(x24)])),  x24 = Concatenate([x, repeat(expand_dims(s, 1), T)])
 * x: (B,T,H,W,CX), s: (B,H,W,CS) = ConvBlock_aux's output, y: (B,T,H,W,CO), ReLU after both 1x1 convolutions.  Parameters = the
 * reference layers' own variables: wt / bt = LocalizedConvBlock's TransitionBlock(2) kernel [1,1,CX+CS,2] / bias, wl / bl =
 * its LocallyConnected2D kernel [H,W,2,2] / bias [H,W,2], w / b = TransitionLast's kernel [1,1,CX+CS+2,CO] / bias.  No
 * concatenation and no time repeat is materialised (csrc/graph_ops4.hip).  dl4ds_rec_tail_supported says whether the channel
 * combination is built (and DL4DS_NO_REC_TAIL_FUSION is unset); otherwise the caller builds the separate layers. */
int dl4ds_rec_tail_supported(int CX, int CS, int CO, int* yes);
int dl4ds_graph_rec_tail(dl4ds_graph* g, int x, int s, int wt, int bt, int wl, int bl, int w, int b, int T, int CO, int* out);
int dl4ds_graph_convlstm(dl4ds_graph* g, int in, int wk, int wr, int b, int KS, int F, int T, int relu, int* out);
int dl4ds_graph_gap(dl4ds_graph* g, int in, int* out);
/* DepthwiseConv2D(7, 'same') + bias -- blocks.py:143-144 */
int dl4ds_graph_dwconv(dl4ds_graph* g, int in, int w, int b, int KS, int* out);
/* GlobalAveragePooling3D over the (time, H, W) axes of a time-distributed tensor -- discriminator.py:73-74 */
int dl4ds_graph_gap3d(dl4ds_graph* g, int in, int* out);
/* y[:, i, j, :] = x[:, oy + i*step, ox + j*step, :] (i < Ho, j < Wo): the sub-sampling half of Conv2D(strides=2)
 * (the stride-1 convolution runs on the MFMA kernels) and Cropping2D -- discriminator.py:53-60 */
int dl4ds_graph_slice(dl4ds_graph* g, int in, int oy, int ox, int step, int Ho, int Wo, int* out);
/* Conv2D(KSxKS, d2s*d2s*Cmid filters) [+ depth_to_space(d2s)] immediately followed by Conv2D(1x1, Cout) (+ optional ReLU)
 * evaluated as one convolution with the composed filter; parameters and their gradients stay those of the two layers
 * (w1 (KS,KS,Cin,d2s^2*Cmid), b1, w2 (Cmid,Cout), b2; biases may be -1).  SubpixelConvolutionBlock / ResizeConvolutionBlock
 * + TransitionBlock 'TransitionLast' -- sp_postups.py:172-177,203. */
int dl4ds_graph_conv2d_folded(dl4ds_graph* g, int in, int w1, int b1, int w2, int b2, int KS, int Cmid, int Cout, int relu,
                              int d2s, int* out);
/* ... with the HR auxiliary branch of the model: TransitionLast reads Concatenate([upsampled x, s]) (sp_postups.py:184-203), i.e.
 * w2 has Cmid + C(aux) rows; out = act(composed conv(x) + conv1x1(aux; rows Cmid..) + bias).  `aux` lives on the output grid. */
int dl4ds_graph_conv2d_folded_aux(dl4ds_graph* g, int in, int aux, int w1, int b1, int w2, int b2, int KS, int Cmid, int Cout,
                                  int relu, int d2s, int* out);
/* ZeroPadding2D(((0, Ho - H), (0, Wo - W))) -- PadConcat, blocks.py:639-647 */
int dl4ds_graph_pad(dl4ds_graph* g, int in, int Ho, int Wo, int* out);
int dl4ds_graph_dense(dl4ds_graph* g, int in, int w, int b, int F, int act, int* out);
int dl4ds_graph_dropout(dl4ds_graph* g, int in, float rate, int* out);
/* get_dropout_layer (blocks.py:679-706).  variant: 0 Dropout, 1 GaussianDropout, 2 SpatialDropout2D/3D (spatial_dim);
 * mc != 0: the MC* layers of blocks.py:658-676, active at inference too. */
int dl4ds_graph_dropout_variant(dl4ds_graph* g, int in, float rate, int variant, int mc, int spatial_dim, int* out);
/* noise of dropout op `index` (creation order) for a batch of B samples: the keep mask (variants 0, 2; one entry per
 * element resp. per (frame|sample, channel)) or the multiplicative Gaussian noise (variant 1).  get: the one the last
 * forward pass used; set: used by the NEXT training-mode forward instead of drawing a new one. */
int dl4ds_graph_dropout_count(dl4ds_graph* g, int* n);
int dl4ds_graph_dropout_mask_size(dl4ds_graph* g, int index, int B, size_t* n);
int dl4ds_graph_dropout_get_mask(dl4ds_graph* g, int index, int B, float* dst_host);
int dl4ds_graph_dropout_set_mask(dl4ds_graph* g, int index, int B, const float* src_host);
/* LayerNormalization(axis=-1) (batch == 0; mov_* ignored) / BatchNormalization(axis=-1, momentum=0.99) (batch != 0)
 * as instantiated by blocks.py:63-71,151-159,293-296; relu != 0 fuses the activation that follows the layer. */
int dl4ds_graph_norm(dl4ds_graph* g, int in, int gamma, int beta, int mov_mean, int mov_var, int batch, float eps, int relu,
                     int* out);
int dl4ds_graph_output(dl4ds_graph* g, int tensor_id);
int dl4ds_graph_finalize(dl4ds_graph* g);
int dl4ds_graph_tensor_shape(dl4ds_graph* g, int tensor_id, int shape4[4]);   /* nmul,H,W,C */
/* parameters: flat fp32 arena; param_id -> (offset, n).  model.get_weights()/set_weights() analogue */
int dl4ds_graph_param_count(dl4ds_graph* g, size_t* n_arena, int* n_params);
int dl4ds_graph_param_info(dl4ds_graph* g, int param_id, size_t* offset, size_t* n);
int dl4ds_graph_set_param(dl4ds_graph* g, int param_id, const float* src_host);
int dl4ds_graph_get_param(dl4ds_graph* g, int param_id, float* dst_host);
int dl4ds_graph_get_grad(dl4ds_graph* g, int param_id, float* dst_host);
int dl4ds_graph_arena_ptrs(dl4ds_graph* g, float** w_dev, float** g_dev);
/* model(inputs, training=...) / model.predict -- inference.py:238; cgan.py:597-600.
 * inputs: n_inputs pointers in graph-input order; is_host selects host or device pointers.
 * out: output tensor 0 copied to out (host or device per is_host); synchronous if is_host. */
int dl4ds_graph_forward(dl4ds_graph* g, const float* const* inputs, int n_inputs, int B, int training, int is_host,
                        float* out);
int dl4ds_graph_tensor_ptr(dl4ds_graph* g, int tensor_id, int grad, float** p_dev);
/* which passes of ChannelAttention2D (blocks.py:585-593) the neighbouring convolutions took over for batch size B
 * (pooling from the producer's epilogue, scale in the consumer's loads, dX in the producer's backward loads): a JSON list,
 * one object per attention layer.  Diagnostics / tests; DL4DS_NO_TAIL_FUSION=1 turns the hand-over off. */
int dl4ds_graph_fusion_report(dl4ds_graph* g, int B, char* json_buf, size_t buflen);
/* the gradient buckets of the data-parallel all-reduce (hvd.DistributedOptimizer's fusion buffer, supervised.py:363-365;
 * hvd.DistributedGradientTape, cgan.py:608-611) in LAUNCH order: a JSON list of {bytes, offset, params,
 * final_after_backward_of_op, of_ops} -- bucket k's ncclAllReduce is queued on the communication stream as soon as the
 * backward pass has run forward op number `final_after_backward_of_op` (the tail of the network goes first). */
int dl4ds_graph_bucket_plan(dl4ds_graph* g, char* json_buf, size_t buflen);

/* ---------------------------------------------------------------- training
 * replaces the Keras fit inner step configured by SupervisedTrainer.run (supervised.py:336-353,396-406):
 * forward -> loss -> backward -> [RCCL all-reduce] -> Adam(PiecewiseConstantDecay). */
int dl4ds_trainer_create(dl4ds_graph* g, int loss_kind, float lr0, float lr1, double lr_boundary, float beta1,
                         float beta2, float eps, dl4ds_trainer** tr);
int dl4ds_trainer_destroy(dl4ds_trainer* tr);
/* one optimisation step.  inputs in graph-input order; y_true (B*nmul,H,W,C) of output 0.
 * loss_host: NULL -> fully asynchronous step; else the loss value (forces a stream sync). */
int dl4ds_trainer_step(dl4ds_trainer* tr, const float* const* inputs, int n_inputs, const float* y_true, int B,
                       int is_host, float* loss_host);
/* loss/gradients without the optimiser update (tests; model.evaluate analogue) */
int dl4ds_trainer_loss_and_grads(dl4ds_trainer* tr, const float* const* inputs, int n_inputs, const float* y_true,
                                 int B, int is_host, float* loss_host);
/* model.evaluate (supervised.py:396-409 validation / test loss): inference-mode forward + loss, no gradients, no
 * BatchNormalization moving-average update, dropout inactive unless it is an MC variant */
int dl4ds_trainer_evaluate(dl4ds_trainer* tr, const float* const* inputs, int n_inputs, const float* y_true, int B,
                           int is_host, float* loss_host);
int dl4ds_trainer_get_state(dl4ds_trainer* tr, float* m_host, float* v_host, long* step);
/* restore the Adam slots (arena-sized host arrays) and optimizer.iterations -- resume from a checkpoint
 * (the reference resumes through tf.train.Checkpoint, cgan.py:288-292; supervised.py:322-325 re-uses a trained model) */
int dl4ds_trainer_set_state(dl4ds_trainer* tr, const float* m_host, const float* v_host, long step);
int dl4ds_trainer_last_loss(dl4ds_trainer* tr, float* loss_host);   /* synchronises */

/* CGAN step -- cgan.py:575-639 with generator_loss (:525-553, lambda) and discriminator_loss (:556-572).
 * gen: generator graph (inputs: lr[, static]); disc: discriminator graph (inputs: lr, hr/generated).
 * losses_host[4] = gen_total, gen_gan, gen_px, disc (NULL -> asynchronous).
 * dropout_keep_host: optional 2*B*C keep-mask (real rows first) for the discriminator's Dropout(0.4). */
int dl4ds_cgan_create(dl4ds_graph* gen, dl4ds_graph* disc, int px_loss_kind, float lr, float beta1, float lam,
                      dl4ds_trainer** tr);
/* genlr, dislr = learning_rates (cgan.py:271-278): one Adam(beta_1=0.5) per model, each with its own rate */
int dl4ds_cgan_set_learning_rates(dl4ds_trainer* tr, float gen_lr, float disc_lr);
int dl4ds_cgan_step(dl4ds_trainer* tr, const float* const* gen_inputs, int n_gen_inputs, const float* hr, int B,
                    int is_host, const float* dropout_keep_host, int apply_update, float* losses_host);
/* Adam slots + optimizer.iterations of the generator (which = 0) / discriminator (which = 1) optimiser -- the contents of
 * the reference's tf.train.Checkpoint(generator_optimizer, discriminator_optimizer, generator, discriminator)
 * (cgan.py:288-292) and what load_checkpoint restores (cgan.py:447-522) */
int dl4ds_cgan_get_state(dl4ds_trainer* tr, int which, float* m_host, float* v_host, long* step);
int dl4ds_cgan_set_state(dl4ds_trainer* tr, int which, const float* m_host, const float* v_host, long step);
int dl4ds_cgan_get_disc_grad(dl4ds_trainer* tr, int param_id, float* dst_host);

/* ---------------------------------------------------------------- batch preparation (SURVEY section 8 "next" row f1)
 * One training batch gathered from a DEVICE-resident dataset; replaces the per-sample numpy/cv2 loop of
 * create_batch_hr_lr / create_pair_hr_lr (dataloader.py:297-360, 11-294) and crop_array / resize_array
 * (utils.py:251-401) for interpolation='inter_area' (integer ratio: block mean; re-expansion for 'pin': replication)
 * and no external LR array.
 *   hr_dev [N][H][W][C], pred_dev [N][H][W][P] or NULL, static_dev [H][W][S] or NULL (all float32, device)
 *   idx_host / cy_host / cx_host [B]: first frame of each sample and its crop corner in HR pixels (host ints; the
 *   caller draws them with the RNG calls of the reference's loop so both paths produce the same batch)
 *   T frames per sample (time_window, 1 for spatial models); patch psy x psx HR pixels (== H x W when not cropping)
 *   pin = 0: out_lr [B][T][psy/scale][psx/scale][C+P(+S)]     pin = 1: out_lr [B][T][psy][psx][C+P(+S)]
 *   static_in_lr: append the (block-mean / raw) static variables to lr (spatial models, dataloader.py:261-289)
 *   out_hr [B][T][psy][psx][C], out_static [B][psy][psx][S] (NULL when S == 0).  Asynchronous on the library stream. */
int dl4ds_batch_prepare(const float* hr_dev, const float* pred_dev, const float* static_dev, const int* idx_host,
                        const int* cy_host, const int* cx_host, float* out_lr_dev, float* out_hr_dev,
                        float* out_static_dev, int H, int W, int C, int P, int S, int T, int B, int scale, int psy,
                        int psx, int pin, int static_in_lr);

/* The same for every interpolation of resize_array (utils.py:369-381: cv2 INTER_NEAREST / INTER_CUBIC / INTER_LINEAR /
 * INTER_AREA / INTER_LANCZOS4).  cv2.resize is separable; one axis of it is a device table of k (source index, weight)
 * taps per output row / column, [n_out][k], built once by the caller from cv2's coefficients.  Each table argument points
 * at two axes {y, x}:
 *   dn_patch : resize of a psy x psx PATCH to (psy/scale) x (psx/scale), indices relative to the patch  (post-upsampling:
 *              the HR crop and the cropped static variables, dataloader.py:141-205,261-289)
 *   dn_field : resize of the whole H x W field to (H/scale) x (W/scale)  (predictors, dataloader.py:150-160; 'pin')
 *   up_field : resize of the (H/scale) x (W/scale) field back to H x W   ('pin', dataloader.py:94-112)
 * scratch_dev: 'pin' only, [B][T][H/scale][W/scale][C+P] floats.  Unused tables may be NULL. */
typedef struct dl4ds_tap_axis { const int* idx; const float* wt; int k; } dl4ds_tap_axis;
int dl4ds_batch_prepare_taps(const float* hr_dev, const float* pred_dev, const float* static_dev, const int* idx_host,
                             const int* cy_host, const int* cx_host, float* out_lr_dev, float* out_hr_dev,
                             float* out_static_dev, float* scratch_dev, int H, int W, int C, int P, int S, int T, int B,
                             int scale, int psy, int psx, int pin, int static_in_lr, const dl4ds_tap_axis* dn_patch,
                             const dl4ds_tap_axis* dn_field, const dl4ds_tap_axis* up_field);

/* The gather pass of the two entries above as a primitive, for the input forms of create_pair_hr_lr they do not cover
 * (dataloader.py:72-73,92-96,149-163,193-200: a caller-supplied LR array, predictors already on the LR grid; utils.py:369-381:
 * cv2.INTER_AREA between grids whose ratio is not an integer):
 *   out[b][t][oy][ox][:] = concat over the groups g of
 *     raw              : src_g[frame][cy_b / row_div + oy][cx_b / row_div + ox][:]      (row_div 0: the corner as given)
 *     origin_from_crop : sum_k wy[oy][ky] wx[ox][kx] src_g[frame][cy_b + iy[oy][ky]][cx_b + ix[ox][kx]][:]  (a PATCH was resized)
 *     row_div > 0      : the same with table rows oy + cy_b / row_div, ox + cx_b / row_div and no origin (the FIELD was resized,
 *                        then cropped on the output grid);   otherwise rows oy, ox, no crop
 *   frame = idx[b] + t (frames 0: dataset), b T + t (frames 1: a batch-local scratch of an earlier pass) or 0 (frames 2: one image).
 * <= 3 groups; idx / cy / cx are host lists of B ints (cy / cx NULL: no crop; idx NULL only without dataset-indexed groups).
 * Asynchronous on the library stream. */
typedef struct dl4ds_gather_group {
    const float* src_dev; int channels; int frames; int src_h; int src_w; int raw; int origin_from_crop; int row_div;
    dl4ds_tap_axis taps[2];
} dl4ds_gather_group;
int dl4ds_batch_gather(const dl4ds_gather_group* groups, int n_groups, const int* idx_host, const int* cy_host,
                       const int* cx_host, float* out_dev, int out_h, int out_w, int T, int B);

/* ---------------------------------------------------------------- data parallelism (RCCL over xGMI)
 * replaces Horovod: hvd.init/rank/size (base.py:97-107), DistributedOptimizer / DistributedGradientTape
 * gradient averaging (supervised.py:365; cgan.py:608-611), broadcast of variables + optimiser slots from
 * rank 0 (supervised.py:369; cgan.py:633-637). */
int dl4ds_dist_unique_id(char id128[128]);                       /* rank 0: ncclGetUniqueId */
int dl4ds_dist_init(int rank, int world, const char id128[128]); /* ncclCommInitRank on the current device */
int dl4ds_dist_world(int* rank, int* world);
int dl4ds_dist_broadcast_trainer(dl4ds_trainer* tr, int root);   /* params + Adam m,v + step */
int dl4ds_dist_allreduce_sum(float* buf_dev, size_t n);          /* on the library stream */
int dl4ds_dist_finalize(void);
/* Fail-safe: the launcher's WORLD_SIZE (1 when unset or DL4DS_ALLOW_UNSYNCED=1).  dl4ds_trainer_step, dl4ds_cgan_step and
 * dl4ds_dist_broadcast_trainer return an error -- they do NOT fall back to local training -- when it is > 1 and no
 * communicator of that size exists (hvd.init() is unconditional in the reference, base.py:97-107). */
int dl4ds_dist_expected_world(int* world);
/* what RCCL reports for the communicator: ncclCommCount / ncclCommUserRank / ncclCommCuDevice (nranks 0: none) */
int dl4ds_dist_comm_info(int* nranks, int* rank, int* device);
/* in-place reduction of n <= 1024 host floats across the ranks; op: 0 sum, 1 max, 2 min; synchronous; identity without a
 * communicator.  Validation / test losses and the early-stopping decision (hvd.callbacks.MetricAverageCallback,
 * supervised.py:366-368), max-over-ranks timing. */
int dl4ds_dist_allreduce_host(float* values_host, int n, int op);
int dl4ds_dist_barrier(void);                                    /* stream sync + a 1-float all-reduce */

#ifdef __cplusplus
}
#endif
#endif /* DL4DS_HIP_H */
