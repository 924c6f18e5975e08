// Entry points whose kernels are not written yet: they fail loudly (no silent fallback).
#include "graph.h"
struct CganTrainer { int dummy; };
#define NOT_YET(name) throw Dl4dsError(std::string(name) + ": not implemented yet in libdl4ds_hip")

size_t conv2d_transpose_workspace_bytes(const TView&, const TView&, int, int) { return 0; }
void conv2d_transpose_forward(hipStream_t, const TView&, const float*, int, int, const TView&, int, float*, size_t) { NOT_YET("conv2d_transpose_forward"); }
void conv2d_transpose_dgrad(hipStream_t, const TView&, const float*, int, int, const TView&, int, float*, size_t) { NOT_YET("conv2d_transpose_dgrad"); }
void conv2d_transpose_wgrad(hipStream_t, const TView&, const TView&, int, int, float*, int, float*, size_t) { NOT_YET("conv2d_transpose_wgrad"); }
int g_conv2d_transpose(Graph&, int, int, int, int, int, int) { NOT_YET("graph conv2d_transpose"); }
int g_convlstm(Graph&, int, int, int, int, int, int, int, int) { NOT_YET("graph convlstm"); }
int g_gap(Graph&, int, int) { NOT_YET("graph gap"); }
int g_dense(Graph&, int, int, int, int, int) { NOT_YET("graph dense"); }
int g_dropout(Graph&, int, float) { NOT_YET("graph dropout"); }
size_t dssim_workspace_bytes(int, int, int, int) { return 0; }
void dssim_forward_backward(hipStream_t, const float*, const float*, float*, int, int, int, int, float, float*, int, float*, size_t) { NOT_YET("dssim"); }
CganTrainer* cgan_create(Graph*, Graph*, int, float, float, float) { NOT_YET("cgan"); }
void cgan_destroy(CganTrainer*) {}
void cgan_step(CganTrainer&, const float* const*, int, const float*, int, bool, const float*, bool, float*) { NOT_YET("cgan"); }
Trainer* cgan_disc_trainer(CganTrainer*) { NOT_YET("cgan"); }
Trainer* cgan_gen_trainer(CganTrainer*) { NOT_YET("cgan"); }
