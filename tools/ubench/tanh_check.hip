// accuracy of the ConvLSTM kernels' tanh on the device: tanh_fast (v_exp_f32 + v_rcp_f32) against libm's tanhf and the fp64 value
// hipcc --offload-arch=gfx950 -O3 tools/ubench/tanh_check.hip -o tools/ubench/tanh_check && tools/ubench/tanh_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__device__ __forceinline__ float tanh_small(float x) {
    const float x2 = x * x;
    return x * fmaf(x2, fmaf(x2, fmaf(x2, -17.f / 315.f, 2.f / 15.f), -1.f / 3.f), 1.f);
}
__device__ __forceinline__ float tanh_fast(float x) {              // (convlstm_seq.hip, round 6)
    const float ax = fabsf(x);
    const float dl = __fdividef(2.f, __expf(2.f * ax) + 1.f);
    return ax < .25f ? tanh_small(x) : copysignf(1.f - dl, x);
}
__device__ __forceinline__ float tanh_old(float x) { return 1.f - __fdividef(2.f, __expf(2.f * x) + 1.f); }
__global__ void k(const float* x, float* a, float* b, float* c, float* e, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { a[i] = tanh_fast(x[i]); b[i] = tanh_old(x[i]); c[i] = tanhf(x[i]); e[i] = __expf(2.f * fabsf(x[i])); }
}
int main() {
    const int n = 1 << 20;
    std::vector<float> x(n), a(n), b(n), c(n), e(n);
    for (int i = 0; i < n; ++i) x[i] = -12.f + 24.f * i / n;
    float *dx, *da, *db, *dc, *de;
    hipMalloc(&dx, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&de, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, da, db, dc, de, n);
    hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost); hipMemcpy(e.data(), de, n * 4, hipMemcpyDeviceToHost);
    // relative error around zero: the round-5 form is a staircase of 6e-8 steps there
    for (float hi = 1e-7f; hi < 1.f; hi *= 10.f) {
        double ra = 0, rb = 0, rc = 0;
        for (int i = 0; i < n; ++i) {
            const float xx = hi * (2.f * i / n - 1.f);
            (void)xx;
        }
        std::vector<float> xs(n);
        for (int i = 0; i < n; ++i) xs[i] = hi * (0.1f + 0.9f * i / n) * ((i & 1) ? 1.f : -1.f);
        hipMemcpy(dx, xs.data(), n * 4, hipMemcpyHostToDevice);
        k<<<n / 256, 256>>>(dx, da, db, dc, de, n);
        hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
        for (int i = 0; i < n; ++i) {
            const double t = std::tanh((double)xs[i]);
            ra = std::fmax(ra, std::fabs(a[i] / t - 1)); rb = std::fmax(rb, std::fabs(b[i] / t - 1)); rc = std::fmax(rc, std::fabs(c[i] / t - 1));
        }
        printf("|x| in [%.0e, %.0e]: max RELATIVE error of tanh: round-6 form %.2e, round-5 form %.2e, tanhf %.2e\n", 0.1 * hi, (double)hi, ra, rb, rc);
    }
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, da, db, dc, de, n);
    hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost); hipMemcpy(e.data(), de, n * 4, hipMemcpyDeviceToHost);
    for (float lo = -12; lo < 12; lo += 1) {
        double ma = 0, mb = 0, mc = 0, me = 0, sa = 0, sb = 0, sc = 0, st = 0;
        for (int i = 0; i < n; ++i) {
            if (x[i] < lo || x[i] >= lo + 1) continue;
            const double t = std::tanh((double)x[i]);
            ma = std::fmax(ma, std::fabs(a[i] - t)); mb = std::fmax(mb, std::fabs(b[i] - t)); mc = std::fmax(mc, std::fabs(c[i] - t));
            me = std::fmax(me, std::fabs(e[i] / std::exp(2.0 * std::fabs((double)x[i])) - 1.0));
            sa += 1.0 - (double)a[i] * a[i]; sb += 1.0 - (double)b[i] * b[i]; sc += 1.0 - (double)c[i] * c[i]; st += 1.0 - t * t;
        }
        printf("[%4.0f,%4.0f) max abs err: fast %.2e old %.2e tanhf %.2e | __expf rel err %.2e | sum(1-t^2)/true: fast %.4f old %.4f tanhf %.4f\n", lo, lo + 1, ma, mb,
               mc, me, sa / st, sb / st, sc / st);
    }
    return 0;
}
