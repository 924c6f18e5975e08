// The instruction mix of a software-pipelined Winograd iteration inside ONE wave per SIMD (gfx950): 36 fp32 MFMAs (three
// accumulator chains, as the 3 cout blocks of a k-loop) interleaved with NP v_pk_add_f32, NR ds_read_b128 and NW ds_write_b128
// of the same wave.  The question: what do the transforms cost the matrix pipe when they are woven into the MFMA stream of
// the wave itself instead of running in the other workgroup's wave (where each vector instruction waits for an MFMA slot)?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_mix.hip -o /tmp/mfma_mix && /tmp/mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NP, int NR, int NW, int WAVES>
__global__ void __launch_bounds__(256, WAVES) k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[256 * 4 * 8];
    f32x4 acc[3];
    for (int i = 0; i < 3; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float a[12], b[4];
    for (int i = 0; i < 12; ++i) a[i] = threadIdx.x * 0.001f + i;
    for (int j = 0; j < 4; ++j) b[j] = threadIdx.x * 0.002f + j;
    f32x2 p[8];
    for (int i = 0; i < 8; ++i) p[i] = (f32x2){(float)i, (float)threadIdx.x};
    f32x4 wv[4];
    for (int i = 0; i < 4; ++i) wv[i] = (f32x4){(float)i, 1.f, 2.f, (float)threadIdx.x};
    f32x4 rd[4];
    for (int i = 0; i < 4; ++i) rd[i] = (f32x4){0, 0, 0, 0};
    float* my = lds + threadIdx.x * 4;
    for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(my + i * 1024) = (f32x4){1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 36; ++m) {
            acc[m % 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m % 12], b[m & 3], acc[m % 3], 0, 0, 0);
            if (m * NP / 36 != (m + 1) * NP / 36 || (NP > 36 && true)) {
#pragma unroll
                for (int r = 0; r < (NP + 35) / 36; ++r) {
                    const int q = (m + r) & 7;
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[q]) : "v"(p[(q + 3) & 7]));
                }
            }
            if (m * NR / 36 != (m + 1) * NR / 36) {
                const int q = (m * NR / 36) & 3;
                asm volatile("ds_read_b128 %0, %1" : "=v"(rd[q]) : "v"((unsigned)(size_t)(my + q * 1024)) : "memory");
            }
            if (m * NW / 36 != (m + 1) * NW / 36) {
                const int q = (m * NW / 36) & 3;
                asm volatile("ds_write_b128 %0, %1" :: "v"((unsigned)(size_t)(my + (4 + q) * 1024)), "v"(wv[q]) : "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = 0;
    for (int i = 0; i < 3; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1];
    for (int i = 0; i < 4; ++i) s += rd[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NP, int NR, int NW, int WAVES>
void run(float* out) {
    const int iters = 20000, grid = 256 * WAVES;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NP, NR, NW, WAVES>), dim3(grid), dim3(256), 0, 0, out, 2000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NP, NR, NW, WAVES>), dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tf = (double)grid * 4 * iters * 36 * 2048.0 / ms / 1e9;
    printf("%d wave/SIMD  36 MFMA + %2d pk_add + %2d ds_read_b128 + %2d ds_write_b128: %7.3f ms  %6.1f TFLOP/s  (%.2f of 157.3)\n",
           WAVES, NP, NR, NW, ms, tf, tf / 157.3);
}

int main() {
    float* out;
    hipMalloc(&out, 512 * 256 * 4);
    run<0, 0, 0, 1>(out);
    run<16, 0, 0, 1>(out);
    run<36, 0, 0, 1>(out);
    run<72, 0, 0, 1>(out);
    run<0, 8, 0, 1>(out);
    run<0, 0, 8, 1>(out);
    run<16, 4, 4, 1>(out);
    run<36, 8, 8, 1>(out);
    run<36, 12, 12, 1>(out);
    run<48, 12, 12, 1>(out);
    run<36, 8, 8, 2>(out);
    run<72, 12, 12, 2>(out);
    return 0;
}
