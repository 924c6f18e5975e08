// Does ONE wave per SIMD sustain the MFMA rate with the accumulator / operand pattern of conv_wgrad_rows
// (27 accumulator tiles, operands fx[6][3] x z[4], order ky,q,kx,i)?   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ORDER>
__global__ void __launch_bounds__(256, 2) k_rows(float* out, int iters) {
    f32x4 acc[9][3];
    for (int t = 0; t < 9; ++t) for (int i = 0; i < 3; ++i) acc[t][i] = (f32x4){0, 0, 0, 0};
    float fx[6][3], z[3][4];
    for (int c = 0; c < 6; ++c) for (int i = 0; i < 3; ++i) fx[c][i] = threadIdx.x * 0.001f + c + 7 * i;
    for (int r = 0; r < 3; ++r) for (int q = 0; q < 4; ++q) z[r][q] = threadIdx.x * 0.002f + q + 5 * r;
    for (int it = 0; it < iters; ++it) {
        if (ORDER == 0) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int i = 0; i < 3; ++i)
                            acc[ky * 3 + kx][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fx[q + kx][i], z[ky][q], acc[ky * 3 + kx][i], 0, 0, 0);
        } else {
            // all 27 accumulators between two uses of the same one
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int i = 0; i < 3; ++i)
                            acc[ky * 3 + kx][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fx[q + kx][i], z[ky][q], acc[ky * 3 + kx][i], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0;
    for (int t = 0; t < 9; ++t) for (int i = 0; i < 3; ++i) s += acc[t][i][0] + acc[t][i][1] + acc[t][i][2] + acc[t][i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F>
float timeit(F f) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) f();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main() {
    float* out;
    if (hipMalloc(&out, 4 * 256 * 2048 * 4) != hipSuccess) return 1;
    const int iters = 500;
    for (int bpc = 1; bpc <= 2; ++bpc) {
        const int grid = 256 * bpc;
        const double fl = (double)grid * 4 * iters * 108 * 2048.0;
        float ms = timeit([&] { hipLaunchKernelGGL((k_rows<0>), dim3(grid), dim3(256), 0, 0, out, iters); });
        printf("order ky,q,kx,i  %d wave/SIMD: %.3f ms  %.1f TFLOP/s\n", bpc, ms, fl / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL((k_rows<1>), dim3(grid), dim3(256), 0, 0, out, iters); });
        printf("order q,ky,kx,i  %d wave/SIMD: %.3f ms  %.1f TFLOP/s\n", bpc, ms, fl / ms / 1e9);
    }
    return 0;
}
