// MFMA f32 16x16x4 issue-rate microbenchmarks (gfx950): register operands vs LDS-fed operands, 1 vs 2 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MT, int NT>
__global__ void __launch_bounds__(256, 2) k_reg(float* out, int iters) {
    f32x4 acc[MT][NT];
    for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    float a[MT], b[NT];
    for (int i = 0; i < MT; ++i) a[i] = threadIdx.x * 0.001f + i;
    for (int j = 0; j < NT; ++j) b[j] = threadIdx.x * 0.002f + j;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// operands from LDS each step with the same prefetch structure as conv_igemm
template <int MT, int NT>
__global__ void __launch_bounds__(256, 2) k_lds(float* out, int iters, int P, int NP) {
    extern __shared__ float smem[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) smem[i] = i * 0.0001f;
    __syncthreads();
    const int lane = threadIdx.x & 63, l15 = lane & 15, lq = lane >> 4, wave = threadIdx.x >> 6;
    f32x4 acc[MT][NT];
    for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    int a_base[MT];
    for (int i = 0; i < MT; ++i) a_base[i] = ((wave * MT + i) * 18 + l15) * P + lq;
    const float* ap = smem;
    const float* bp = smem + 8192 + lq * NP + l15;
    float av[MT], bv[NT];
    for (int i = 0; i < MT; ++i) av[i] = ap[a_base[i] & 8191];
    for (int j = 0; j < NT; ++j) bv[j] = bp[j * 16];
    for (int it = 0; it < iters; ++it) {
        for (int kk = 0; kk < 12; ++kk) {
            const int kn = (kk + 1 < 12) ? kk + 1 : kk;
            float an[MT], bn[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) an[i] = ap[(a_base[i] + kn * 4) & 8191];
#pragma unroll
            for (int j = 0; j < NT; ++j) bn[j] = bp[(kn * 4 * NP + j * 16) & 8191];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MT; ++i) av[i] = an[i];
#pragma unroll
            for (int j = 0; j < NT; ++j) bv[j] = bn[j];
        }
    }
    float s = 0;
    for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F>
float timeit(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main() {
    float* out;
    CK(hipMalloc(&out, 4 * 256 * 2048 * 4));
    const int iters_reg = 2000, iters_lds = 160;
    for (int blocks_per_cu = 1; blocks_per_cu <= 2; ++blocks_per_cu) {
        const int grid = 256 * blocks_per_cu;
        {
            float ms = timeit([&] { hipLaunchKernelGGL((k_reg<4, 6>), dim3(grid), dim3(256), 0, 0, out, iters_reg); });
            double fl = (double)grid * 4 * iters_reg * 24 * 2048.0;
            printf("reg  4x6 acc, %d block/CU: %.3f ms  %.1f TFLOP/s\n", blocks_per_cu, ms, fl / ms / 1e9);
        }
        {
            float ms = timeit([&] { hipLaunchKernelGGL((k_reg<2, 2>), dim3(grid), dim3(256), 0, 0, out, iters_reg * 6); });
            double fl = (double)grid * 4 * iters_reg * 6 * 4 * 2048.0;
            printf("reg  2x2 acc, %d block/CU: %.3f ms  %.1f TFLOP/s\n", blocks_per_cu, ms, fl / ms / 1e9);
        }
        {
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds<4, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
            float ms = timeit([&] { hipLaunchKernelGGL((k_lds<4, 6>), dim3(grid), dim3(256), 72 * 1024, 0, out, iters_lds, 50, 208); });
            double fl = (double)grid * 4 * iters_lds * 12 * 24 * 2048.0;
            printf("lds  4x6 acc, %d block/CU: %.3f ms  %.1f TFLOP/s\n", blocks_per_cu, ms, fl / ms / 1e9);
        }
    }
    return 0;
}
