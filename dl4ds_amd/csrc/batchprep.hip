// dl4ds_amd -- on-device batch preparation ("next" row f1): the step immediately BEFORE the train step.
//
// Replaces the per-sample Python/cv2 loop of dl4ds/dataloader.py:11-360 (create_pair_hr_lr / create_batch_hr_lr) and
// dl4ds/utils.py:251-401 (crop_array / resize_array) for the default path the trainers take: 'inter_area'
// interpolation (cv2.INTER_AREA at an integer ratio == block mean; at an integer up-scaling ratio == pixel
// replication), no externally supplied LR array.  The whole dataset (HR fields, optional predictor fields, optional
// static variables) lives in HBM; a batch is produced by two gather kernels from (sample index, crop corner) lists
// drawn on the host with the same RNG calls as the numpy port, so both paths yield the same batches:
//   post-upsampling :  lr[b,t,y,x,:] = [ blockmean_s(hr) | blockmean_s(predictors) | blockmean_s(static) ]   (LR grid)
//   pre-upsampling  :  lr[b,t,Y,X,:] = [ replicate(blockmean_s(hr)) | replicate(blockmean_s(predictors)) | static ]
//   hr[b,t,Y,X,:]   =  crop of the HR field,   static_hr[b,Y,X,:] = crop of the static variables
// (static variables are appended to lr only for spatial models, dataloader.py:261-289 -> `static_in_lr`).
// Pure HBM streaming: every HR pixel of the crop is read once for the block means and once for the HR copy.
#include "ops.h"
#include "prof.h"
#include <algorithm>

namespace {

struct PrepParams {
    const float* hr;        // [N][H][W][C]
    const float* pred;      // [N][H][W][P] or null
    const float* stat;      // [H][W][S] or null
    const int* idx;         // [B] first frame of each sample
    const int* cy;          // [B] crop corner in HR pixels
    const int* cx;
    float* out_lr;          // post: [B][T][ps/s][ps/s][CL]   pin: [B][T][ps][ps][CL]
    float* out_hr;          // [B][T][ps][ps][C]
    float* out_stat;        // [B][ps][ps][S] or null
    int H, W, C, P, S, T, B;
    int scale, ps;          // ps = HR patch size (== H == W when not cropping; H, W may differ -> psy, psx)
    int psy, psx;
    int pin;                // 1: LR is re-expanded to the HR grid
    int static_in_lr;       // append (block-mean / raw) static variables to lr
};

// mean over the scale x scale block whose top-left PIXEL is (py, px) of channel c of a [H][W][Cn] image
__device__ __forceinline__ float block_mean(const float* __restrict__ img, int W, int Cn, int c, int py, int px, int s) {
    float acc = 0.f;
    const float* p = img + ((size_t)py * W + (size_t)px) * Cn + c;
    for (int dy = 0; dy < s; ++dy)
        for (int dx = 0; dx < s; ++dx) acc += p[((size_t)dy * W + dx) * Cn];
    return acc / (float)(s * s);
}

__global__ void prep_lr_kernel(const PrepParams a) {
    const int CL = a.C + a.P + (a.static_in_lr ? a.S : 0);
    const int oy_n = a.pin ? a.psy : a.psy / a.scale, ox_n = a.pin ? a.psx : a.psx / a.scale;
    const size_t total = (size_t)a.B * a.T * oy_n * ox_n * CL;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % CL);
        size_t r = e / CL;
        const int ox = (int)(r % ox_n); r /= ox_n;
        const int oy = (int)(r % oy_n); r /= oy_n;
        const int t = (int)(r % a.T);
        const int b = (int)(r / a.T);
        // absolute HR position of this output pixel (pin) or of its block's corner (post)
        const int Y = a.pin ? a.cy[b] + oy : a.cy[b] + oy * a.scale;
        const int X = a.pin ? a.cx[b] + ox : a.cx[b] + ox * a.scale;
        // 'pin': the WHOLE field is coarsened and re-expanded before the crop (dataloader.py:102-106), so the block is
        // the one of the field's own LR grid that contains the pixel; post-upsampling: the crop is taken first (at any
        // pixel) and the PATCH is coarsened (dataloader.py:201-205), so blocks are aligned with the crop corner
        const int by = a.pin ? (Y / a.scale) * a.scale : Y, bx = a.pin ? (X / a.scale) * a.scale : X;
        const size_t frame = (size_t)(a.idx[b] + t) * a.H * a.W;
        float v;
        if (c < a.C) {
            v = block_mean(a.hr + frame * a.C, a.W, a.C, c, by, bx, a.scale);
        } else if (c < a.C + a.P) {
            v = block_mean(a.pred + frame * a.P, a.W, a.P, c - a.C, by, bx, a.scale);
        } else if (a.pin) {
            v = a.stat[((size_t)Y * a.W + X) * a.S + (c - a.C - a.P)];
        } else {
            v = block_mean(a.stat, a.W, a.S, c - a.C - a.P, by, bx, a.scale);
        }
        a.out_lr[e] = v;
    }
}

__global__ void prep_hr_kernel(const PrepParams a) {
    const size_t n_hr = (size_t)a.B * a.T * a.psy * a.psx * a.C;
    const size_t n_st = a.out_stat ? (size_t)a.B * a.psy * a.psx * a.S : 0;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_hr + n_st; e += (size_t)gridDim.x * blockDim.x) {
        if (e < n_hr) {
            const int c = (int)(e % a.C);
            size_t r = e / a.C;
            const int x = (int)(r % a.psx); r /= a.psx;
            const int y = (int)(r % a.psy); r /= a.psy;
            const int t = (int)(r % a.T);
            const int b = (int)(r / a.T);
            a.out_hr[e] = a.hr[(((size_t)(a.idx[b] + t) * a.H + a.cy[b] + y) * a.W + a.cx[b] + x) * a.C + c];
        } else {
            const size_t q = e - n_hr;
            const int c = (int)(q % a.S);
            size_t r = q / a.S;
            const int x = (int)(r % a.psx); r /= a.psx;
            const int y = (int)(r % a.psy);
            const int b = (int)(r / a.psy);
            a.out_stat[q] = a.stat[((size_t)(a.cy[b] + y) * a.W + a.cx[b] + x) * a.S + c];
        }
    }
}

}  // namespace

void batch_prepare(hipStream_t s, const float* hr, const float* pred, const float* stat, const int* idx, const int* cy,
                   const int* cx, float* out_lr, float* out_hr, float* out_stat, int H, int W, int C, int P, int S, int T,
                   int B, int scale, int psy, int psx, int pin, int static_in_lr) {
    DL4DS_REQUIRE(B > 0 && T > 0 && C > 0 && scale >= 1, "batch_prepare: bad sizes");
    DL4DS_REQUIRE(H % scale == 0 && W % scale == 0, "batch_prepare: field size must be divisible by scale (block mean)");
    DL4DS_REQUIRE(psy <= H && psx <= W, "batch_prepare: patch larger than the field");
    DL4DS_REQUIRE(pin || (psy % scale == 0 && psx % scale == 0), "batch_prepare: patch size must be divisible by scale");
    DL4DS_REQUIRE((P == 0) == (pred == nullptr) && (S == 0) == (stat == nullptr), "batch_prepare: predictor/static pointers");
    PrepParams a;
    a.hr = hr; a.pred = pred; a.stat = stat; a.idx = idx; a.cy = cy; a.cx = cx;
    a.out_lr = out_lr; a.out_hr = out_hr; a.out_stat = S ? out_stat : nullptr;
    a.H = H; a.W = W; a.C = C; a.P = P; a.S = S; a.T = T; a.B = B; a.scale = scale; a.ps = psy; a.psy = psy; a.psx = psx;
    a.pin = pin; a.static_in_lr = (static_in_lr && S) ? 1 : 0;
    const int CL = C + P + (a.static_in_lr ? S : 0);
    const size_t n_lr = (size_t)B * T * (pin ? psy : psy / scale) * (pin ? psx : psx / scale) * CL;
    const size_t n_hr = (size_t)B * T * psy * psx * C + (S ? (size_t)B * psy * psx * S : 0);
    const double crop_bytes = 4.0 * B * T * (double)psy * psx * (C + P);
    {
        ProfScope ps(s, "batch_prepare_lr", 0.0, crop_bytes + 4.0 * n_lr);
        const int blocks = (int)std::max<size_t>(1, std::min<size_t>(cdivz(n_lr, 256), 16384));
        hipLaunchKernelGGL(prep_lr_kernel, dim3(blocks), dim3(256), 0, s, a);
        HIP_CHECK(hipGetLastError());
    }
    {
        ProfScope ps(s, "batch_prepare_hr", 0.0, 8.0 * n_hr);
        const int blocks = (int)std::max<size_t>(1, std::min<size_t>(cdivz(n_hr, 256), 16384));
        hipLaunchKernelGGL(prep_hr_kernel, dim3(blocks), dim3(256), 0, s, a);
        HIP_CHECK(hipGetLastError());
    }
}
