// Issue rate of v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4 outer products per instruction: 512 FLOP) against
// v_mfma_f32_16x16x4_f32 (2048 FLOP) on gfx950, from registers, with NACC independent accumulators per wave and one or two
// waves per SIMD.  The question behind it: a 3x3 layer with 8 output channels fills only half the rows of a 16x16 MFMA,
// the 4x4 form has no idle rows -- is its FLOP rate the same?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_4x4.hip -o /tmp/mfma_4x4 && /tmp/mfma_4x4
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool SMALL>
__global__ void __launch_bounds__(256, 2) k(float* out, int iters) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = threadIdx.x * 0.002f + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (SMALL) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[r], b[(r + i) & 3], acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], b[(r + i) & 3], acc[i], 0, 0, 0);
            }
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, bool SMALL>
void run(float* out, int bpc) {
    const int grid = 256 * bpc, iters = SMALL ? 40000 : 10000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, SMALL>), dim3(grid), dim3(256), 0, 0, out, 200);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, SMALL>), dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)grid * 4 * iters * 4 * NACC;
    const double fl = n * (SMALL ? 512.0 : 2048.0);
    // cycles per instruction per SIMD at a nominal 2.4 GHz
    const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 4 * NACC * bpc);
    printf("%s  %d acc  %d wave(s)/SIMD: %8.3f ms  %6.1f TFLOP/s  %.1f cycles/instr/SIMD @2.4GHz\n",
           SMALL ? "4x4x1_16B " : "16x16x4   ", NACC, bpc, ms, fl / ms / 1e9, cyc);
}

int main() {
    float* out;
    hipMalloc(&out, 512 * 256 * 4);
    for (int bpc = 1; bpc <= 2; ++bpc) {
        run<1, true>(out, bpc); run<2, true>(out, bpc); run<3, true>(out, bpc); run<6, true>(out, bpc); run<12, true>(out, bpc);
        run<1, false>(out, bpc); run<2, false>(out, bpc); run<6, false>(out, bpc);
    }
    return 0;
}
