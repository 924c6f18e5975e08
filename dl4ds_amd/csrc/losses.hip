// Pixel losses of dl4ds/losses.py (mae :5-11, mse :14-20, dssim :23-55 and the 0.8/0.2, 0.6/0.2/0.2
// mixes :58-89) and Keras BinaryCrossentropy(from_logits=False) (cgan.py:546-549,567-571), each fused
// with its gradient w.r.t. the prediction.  Wave reductions -> one partial per block -> finish kernel.
#include "ops.h"
#include "prof.h"
#include <algorithm>

namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) red[wv] = v;
    __syncthreads();
    float s = 0.f;
    if (threadIdx.x == 0) for (int k = 0; k < (int)(blockDim.x >> 6); ++k) s += red[k];
    __syncthreads();
    return s;   // valid in thread 0
}

// partial[2*b] = sum |d| ; partial[2*b+1] = sum d^2 ; dpred (+)= ga*sign(d) + gs*2*d
__global__ void __launch_bounds__(256) pixel_loss_kernel(const float* __restrict__ t, const float* __restrict__ p,
                                                         float* __restrict__ dp, size_t n, float ga, float gs,
                                                         int accumulate, float* __restrict__ partial) {
    __shared__ float red[4];
    float sa = 0.f, ss = 0.f;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const float d = p[e] - t[e];
        sa += fabsf(d);
        ss += d * d;
        if (dp) {
            const float sg = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
            const float g = ga * sg + gs * 2.f * d;
            dp[e] = accumulate ? dp[e] + g : g;
        }
    }
    const float a = block_sum(sa, red);
    const float b = block_sum(ss, red);
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = a; partial[2 * blockIdx.x + 1] = b; }
}

// loss_out[0] (+)= wa*mean|d| + ws*mean d^2
__global__ void pixel_loss_finish_kernel(const float* __restrict__ partial, int nb, float wa, float ws, float inv_n,
                                         float* loss_out, int accumulate) {
    // one wave: lane-strided partial sums, then a fixed butterfly (deterministic; the single-thread chain took 53 us)
    float a = 0.f, b = 0.f;
    for (int k = threadIdx.x; k < nb; k += 64) { a += partial[2 * k]; b += partial[2 * k + 1]; }
    a = wave_sum(a);
    b = wave_sum(b);
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float v = wa * a * inv_n + ws * b * inv_n;
        loss_out[0] = accumulate ? loss_out[0] + v : v;
    }
}

__global__ void __launch_bounds__(256) bce_kernel(const float* __restrict__ p, float label, int n, float scale,
                                                  float* loss_out, float* __restrict__ dp, int accumulate_loss) {
    __shared__ float red[4];
    const float eps = 1e-7f;
    float s = 0.f;
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const float pr = p[e];
        const float pc = fminf(fmaxf(pr, eps), 1.f - eps);
        s += -(label * logf(pc) + (1.f - label) * logf(1.f - pc));
        if (dp) {
            // d/dp of the clipped form: zero outside [eps, 1-eps]
            float g = 0.f;
            if (pr >= eps && pr <= 1.f - eps) g = -(label / pc - (1.f - label) / (1.f - pc));
            dp[e] = scale * g / (float)n;
        }
    }
    const float tot = block_sum(s, red);
    if (threadIdx.x == 0) {
        const float v = scale * tot / (float)n;
        loss_out[0] = accumulate_loss ? loss_out[0] + v : v;
    }
}

int loss_blocks(size_t n) { return (int)std::max<size_t>(1, std::min<size_t>(cdivz(n, 256 * 8), 1024)); }

}  // namespace

void dssim_forward_backward(hipStream_t s, const float* y_true, const float* y_pred, float* dpred, int N, int H, int W,
                            int C, float weight, float* loss_out, int accumulate_loss, float* workspace,
                            size_t workspace_bytes);
size_t dssim_workspace_bytes(int N, int H, int W, int C);
void msdssim_forward_backward(hipStream_t s, const float* y_true, const float* y_pred, float* dpred, int N, int H, int W,
                              int C, float weight, float* loss_out, int accumulate_loss, float* workspace,
                              size_t workspace_bytes);
size_t msdssim_workspace_bytes(int N, int H, int W, int C);

static void loss_weights(int kind, float& wd, float& wa, float& ws, float& wm) {
    wd = wa = ws = wm = 0.f;
    switch (kind) {
        case LOSS_MAE: wa = 1.f; break;
        case LOSS_MSE: ws = 1.f; break;
        case LOSS_DSSIM: wd = 1.f; break;
        case LOSS_DSSIM_MAE: wd = 0.8f; wa = 0.2f; break;
        case LOSS_DSSIM_MSE: wd = 0.8f; ws = 0.2f; break;
        case LOSS_DSSIM_MAE_MSE: wd = 0.6f; wa = 0.2f; ws = 0.2f; break;
        case LOSS_MSDSSIM: wm = 1.f; break;                                   // losses.py:92-130
        case LOSS_MSDSSIM_MAE: wm = 0.8f; wa = 0.2f; break;                   // :133-139
        case LOSS_MSDSSIM_MAE_MSE: wm = 0.6f; wa = 0.2f; ws = 0.2f; break;    // :142-149
        default: throw Dl4dsError("unknown loss kind " + std::to_string(kind));
    }
}

size_t loss_workspace_bytes(int kind, int N, int H, int W, int C) {
    float wd, wa, ws, wm;
    loss_weights(kind, wd, wa, ws, wm);
    size_t b = 2 * 1024 * sizeof(float);
    if (wd != 0.f) b += dssim_workspace_bytes(N, H, W, C);
    if (wm != 0.f) b += msdssim_workspace_bytes(N, H, W, C);
    return b;
}

void loss_forward_backward(hipStream_t s, int kind, const float* y_true, const float* y_pred, float* dpred, int N,
                           int H, int W, int C, float scale, float* loss_out, int accumulate, float* workspace,
                           size_t workspace_bytes) {
    float wd, wa, ws, wm;
    loss_weights(kind, wd, wa, ws, wm);
    const size_t n = (size_t)N * H * W * C;
    DL4DS_REQUIRE(workspace_bytes >= loss_workspace_bytes(kind, N, H, W, C), "loss workspace too small");
    const int nb = loss_blocks(n);
    const float inv_n = 1.f / (float)n;
    ProfScope ps(s, "pixel_loss", 0.0, 12.0 * (double)n);
    DL4DS_LAUNCH(pixel_loss_kernel, dim3(nb), dim3(256), 0, s, y_true, y_pred, dpred, n, scale * wa * inv_n,
                       scale * ws * inv_n, accumulate, workspace);
    HIP_CHECK(hipGetLastError());
    DL4DS_LAUNCH(pixel_loss_finish_kernel, dim3(1), dim3(64), 0, s, workspace, nb, scale * wa, scale * ws, inv_n,
                       loss_out, 0);
    HIP_CHECK(hipGetLastError());
    if (wd != 0.f) {
        dssim_forward_backward(s, y_true, y_pred, dpred, N, H, W, C, scale * wd, loss_out, 1, workspace + 2 * 1024,
                               workspace_bytes - 2 * 1024 * sizeof(float));
    }
    if (wm != 0.f) {
        msdssim_forward_backward(s, y_true, y_pred, dpred, N, H, W, C, scale * wm, loss_out, 1, workspace + 2 * 1024,
                                 workspace_bytes - 2 * 1024 * sizeof(float));
    }
}

void bce_forward_backward(hipStream_t s, const float* p, float label, int n, float scale, float* loss_out, float* dp,
                          int accumulate_loss) {
    DL4DS_LAUNCH(bce_kernel, dim3(1), dim3(256), 0, s, p, label, n, scale, loss_out, dp, accumulate_loss);
    HIP_CHECK(hipGetLastError());
}
