// What the fp32 matrix cores sustain when NOTHING else is in the way, and at which shader clock: 256 CUs x 4 SIMDs, one
// or two wavefronts per SIMD issuing v_mfma_f32_16x16x4_f32 from registers (24 independent accumulators), no memory
// traffic at all.  Every wave reads the shader-clock counter (s_memtime) and the constant 100 MHz counter
// (s_memrealtime) around its loop: their ratio is the clock the chip actually ran at under this load.  The datasheet
// peak (157.3 TFLOP/s) assumes 2.4 GHz; under sustained MFMA load the chip is power-managed below that.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_clock.hip -o /tmp/mfma_clock && /tmp/mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256, 2) k(float* out, unsigned long long* clk, int iters) {
    f32x4 acc[4][6];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 6; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    float a[4], b[6];
    for (int i = 0; i < 4; ++i) a[i] = threadIdx.x * 0.001f + i;
    for (int j = 0; j < 6; ++j) b[j] = threadIdx.x * 0.002f + j;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 6; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        clk[2 * w] = c1 - c0;
        clk[2 * w + 1] = w1 - w0;
    }
}

int main() {
    float* out; unsigned long long* clk;
    hipMalloc(&out, 512 * 256 * 4);
    hipMalloc(&clk, 512 * 4 * 2 * 8);
    for (int bpc = 1; bpc <= 2; ++bpc) {
        for (int iters : {2000, 20000, 100000}) {
            const int grid = 256 * bpc;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, clk, 200);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, clk, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(grid * 4 * 2);
            hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
            std::vector<double> mhz;
            for (int w = 0; w < grid * 4; ++w) mhz.push_back(100.0 * (double)h[2 * w] / (double)h[2 * w + 1]);
            std::sort(mhz.begin(), mhz.end());
            const double fl = (double)grid * 4 * iters * 24 * 2048.0;
            printf("%d wave(s)/SIMD, %6d iters: %8.3f ms  %6.1f TFLOP/s  shader clock under load: median %.0f MHz (min %.0f, max %.0f)"
                   "  -> peak at that clock %.1f TFLOP/s\n", bpc, iters, ms, fl / ms / 1e9, mhz[mhz.size() / 2], mhz.front(), mhz.back(),
                   256 * 4 * 64.0 * mhz[mhz.size() / 2] * 1e6 / 1e12);
        }
    }
    return 0;
}
