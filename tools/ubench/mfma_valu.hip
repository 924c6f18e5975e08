// What an ordinary VALU instruction costs a stream of fp32 MFMAs (gfx950).  One or two wavefronts per SIMD issue
// v_mfma_f32_16x16x4_f32 from registers (24 independent accumulators, 24 MFMAs per iteration = 768 matrix-pipe cycles);
// K extra VALU instructions are inserted per iteration, either in the MFMA wave itself ("own") or only in the second
// wave of the SIMD while the first one issues pure MFMAs ("other": every second workgroup runs a VALU-only loop).
// If VALU work ran beside the matrix pipe the MFMA rate would not depend on K.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu.hip -o /tmp/mfma_valu && /tmp/mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int K, int KIND>      // KIND 0: v_add_u32 (independent chain x4), 1: v_lshl_add_u64, 2: v_fma_f32
__global__ void __launch_bounds__(256, 2) k_own(float* out, int iters) {
    f32x4 acc[4][6];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 6; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    float a[4], b[6];
    for (int i = 0; i < 4; ++i) a[i] = threadIdx.x * 0.001f + i;
    for (int j = 0; j < 6; ++j) b[j] = threadIdx.x * 0.002f + j;
    unsigned x[4] = {threadIdx.x, threadIdx.x + 1, threadIdx.x + 2, threadIdx.x + 3};
    unsigned long long y[4] = {threadIdx.x, threadIdx.x + 7ull, threadIdx.x + 9ull, threadIdx.x + 11ull};
    float z[4] = {1.f, 2.f, 3.f, 4.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
                constexpr int idx = 0;
                (void)idx;
                if ((i * 6 + j) < K) {
                    const int q = (i * 6 + j) & 3;
                    if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[q]) : "v"(x[(q + 1) & 3]));
                    else if (KIND == 1) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(y[q]) : "v"(y[(q + 1) & 3]));
                    else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(z[q]) : "v"(z[(q + 1) & 3]));
                }
            }
    }
    float s = (float)(x[0] + x[1] + x[2] + x[3]) + (float)(y[0] + y[1] + y[2] + y[3]) + z[0] + z[1] + z[2] + z[3];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 6; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// workgroups with odd HW slot (HW_ID.tg_id) run a VALU-only loop, even ones pure MFMAs: the two kinds share every SIMD
template <int K>
__global__ void __launch_bounds__(256, 2) k_other(float* out, int iters, unsigned long long* mf_cycles) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    const bool valu_wave = ((hw >> 16) & 1u) != 0;
    if (valu_wave) {
        unsigned x[4] = {threadIdx.x, threadIdx.x + 1, threadIdx.x + 2, threadIdx.x + 3};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < K; ++k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[k & 3]) : "v"(x[(k + 1) & 3]));
            if (K < 24) __builtin_amdgcn_s_sleep(K == 0 ? 12 : (24 - K) / 2);     // keep the loop period comparable
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(x[0] + x[1] + x[2] + x[3]);
        return;
    }
    f32x4 acc[4][6];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 6; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    float a[4], b[6];
    for (int i = 0; i < 4; ++i) a[i] = threadIdx.x * 0.001f + i;
    for (int j = 0; j < 6; ++j) b[j] = threadIdx.x * 0.002f + j;
    const unsigned long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    const unsigned long long c1 = clock64();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 6; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) mf_cycles[blockIdx.x] = c1 - c0;
}

template <class F>
float timeit(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 3;
}

template <int K, int KIND>
void run_own(float* out, int bpc) {
    const int iters = 20000, grid = 256 * bpc;
    float ms = timeit([&] { hipLaunchKernelGGL((k_own<K, KIND>), dim3(grid), dim3(256), 0, 0, out, iters); });
    const double cyc = ms * 1e-3 * 2.4e9 / iters / (bpc == 2 ? 2 : 1);         // cycles per iteration per wave-pair slot (nominal clock)
    printf("own   kind %d  K=%2d VALU per 24 MFMAs, %d wave/SIMD: %7.3f ms  %6.1f TFLOP/s  (%.0f cycles per 24 MFMAs; 768 = matrix pipe alone)\n",
           KIND, K, bpc, ms, (double)grid * 4 * iters * 24 * 2048.0 / ms / 1e9, cyc);
}

template <int K>
void run_other(float* out, unsigned long long* cyc) {
    const int iters = 20000, grid = 512;
    hipMemset(cyc, 0, 512 * 8);
    float ms = timeit([&] { hipLaunchKernelGGL((k_other<K>), dim3(grid), dim3(256), 0, 0, out, iters, cyc); });
    unsigned long long h[512];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double sum = 0; int n = 0;
    for (int i = 0; i < 512; ++i) if (h[i]) { sum += (double)h[i]; ++n; }
    printf("other K=%2d VALU per iteration in the co-resident wave: %7.3f ms, %d MFMA workgroups, %.0f shader cycles per 24 MFMAs (768 = alone)\n",
           K, ms, n, n ? sum / n / iters : 0.0);
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 512 * 256 * 4);
    hipMalloc(&cyc, 512 * 8);
    for (int bpc = 1; bpc <= 2; ++bpc) {
        run_own<0, 0>(out, bpc); run_own<2, 0>(out, bpc); run_own<6, 0>(out, bpc); run_own<12, 0>(out, bpc); run_own<24, 0>(out, bpc);
        run_own<6, 1>(out, bpc); run_own<6, 2>(out, bpc);
    }
    run_other<0>(out, cyc); run_other<6>(out, cyc); run_other<12>(out, cyc); run_other<24>(out, cyc);
    return 0;
}
