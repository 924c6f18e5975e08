// fp32 products on the 16-bit matrix pipe of gfx950: RATE and ACCURACY of split-operand MFMAs (round 5).
//   x = hi + lo (+ lo2), hi = bf16(x), lo = bf16(x - hi), lo2 = bf16(x - hi - lo);  a b ~ ahi bhi + ahi blo + alo bhi  (3 MFMAs, "x3")
//   ... + ahi blo2 + alo2 bhi + alo blo (6 MFMAs, "x6"), accumulated in fp32 by v_mfma_f32_16x16x32_bf16 (16384 FLOP per instruction
//   against 2048 of v_mfma_f32_16x16x4_f32).
// Part 1: issue rate from registers (NACC independent accumulators, one / two waves per SIMD), no operand traffic: the roof of the form.
// Part 2: C = A B for a 16 x 16 tile with K = 288 (9 taps x 32 channels) on N(0,1) data against fp64: max |error| / max |C|.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_split.hip -o /tmp/mfma_split && /tmp/mfma_split
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int MODE>      // MODE 0: f32 16x16x4; 1: bf16 16x16x32
__global__ void __launch_bounds__(256, 2) rate_k(float* out, int iters) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f;
    bf16x8 ha, hb;
    for (int j = 0; j < 8; ++j) { ha[j] = (__bf16)(a + j); hb[j] = (__bf16)(b - j); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, acc[i], 0, 0, 0);
            }
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int MODE>
void rate(float* out, int bpc) {
    const int grid = 256 * bpc, iters = 10000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((rate_k<NACC, MODE>), dim3(grid), dim3(256), 0, 0, out, 200);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((rate_k<NACC, MODE>), dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)grid * 4 * iters * 4 * NACC;
    const double fl = n * (MODE == 0 ? 2048.0 : 16384.0);
    printf("%s  %2d acc  %d wave(s)/SIMD: %8.3f ms  %7.1f TFLOP/s issued   = %6.1f TFLOP/s of fp32 products as x3, %6.1f as x6\n",
           MODE == 0 ? "f32  16x16x4 " : "bf16 16x16x32", NACC, bpc, ms, fl / ms / 1e9, MODE ? fl / ms / 1e9 / 3 : fl / ms / 1e9,
           MODE ? fl / ms / 1e9 / 6 : fl / ms / 1e9);
}

// one wave: C[16][16] = A[16][K] B[K][16]; lane l: row / column l % 16, k-group g = l / 16
template <int MODE>      // 0: fp32 MFMA; 1: bf16 x1; 3: x3; 6: x6
__global__ void acc_k(const float* A, const float* B, float* C, int K) {
    const int l = threadIdx.x, rc = l & 15, g = l >> 4;
    f32x4 c = {0, 0, 0, 0};
    if (MODE == 0) {
        for (int k0 = 0; k0 < K; k0 += 4) c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[rc * K + k0 + g], B[(k0 + g) * 16 + rc], c, 0, 0, 0);
    } else {
        for (int k0 = 0; k0 < K; k0 += 32) {
            bf16x8 ah, al, al2, bh, bl, bl2;
            for (int j = 0; j < 8; ++j) {
                const float a = A[rc * K + k0 + 8 * g + j], b = B[(k0 + 8 * g + j) * 16 + rc];
                ah[j] = (__bf16)a; const float ra = a - (float)ah[j]; al[j] = (__bf16)ra; al2[j] = (__bf16)(ra - (float)al[j]);
                bh[j] = (__bf16)b; const float rb = b - (float)bh[j]; bl[j] = (__bf16)rb; bl2[j] = (__bf16)(rb - (float)bl[j]);
            }
            // small terms first
            if (MODE >= 6) {
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bl, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl2, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al2, bh, c, 0, 0, 0);
            }
            if (MODE >= 3) {
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
            }
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
        }
    }
    for (int r = 0; r < 4; ++r) C[(4 * g + r) * 16 + rc] = c[r];
}

template <int MODE>
void accuracy(const float* dA, const float* dB, float* dC, const std::vector<double>& ref, int K, const char* name) {
    hipLaunchKernelGGL(acc_k<MODE>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
    std::vector<float> c(256);
    hipMemcpy(c.data(), dC, 256 * 4, hipMemcpyDeviceToHost);
    double e = 0, m = 0;
    for (int i = 0; i < 256; ++i) { e = std::fmax(e, std::fabs(c[i] - ref[i])); m = std::fmax(m, std::fabs(ref[i])); }
    printf("%-22s K = %d: max |C - C_fp64| / max |C| = %.3e\n", name, K, e / m);
}

int main() {
    float* out;
    hipMalloc(&out, 512 * 256 * 4);
    printf("---- issue rate (registers only)\n");
    for (int bpc = 1; bpc <= 2; ++bpc) {
        rate<2, 0>(out, bpc); rate<6, 0>(out, bpc);
        rate<1, 1>(out, bpc); rate<2, 1>(out, bpc); rate<4, 1>(out, bpc); rate<6, 1>(out, bpc);
    }
    printf("---- accuracy of one 16 x 16 tile\n");
    for (int K : {288, 1152}) {
        std::mt19937 rng(7);
        std::normal_distribution<float> nd(0.f, 1.f);
        std::vector<float> A(16 * K), B(K * 16);
        for (auto& v : A) v = nd(rng);
        for (auto& v : B) v = nd(rng);
        std::vector<double> ref(256, 0.0);
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * B[k * 16 + j]; ref[i * 16 + j] = s; }
        float *dA, *dB, *dC;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 256 * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        accuracy<0>(dA, dB, dC, ref, K, "fp32 MFMA 16x16x4");
        accuracy<1>(dA, dB, dC, ref, K, "bf16 x1");
        accuracy<3>(dA, dB, dC, ref, K, "bf16 x3 (hi/lo)");
        accuracy<6>(dA, dB, dC, ref, K, "bf16 x6 (hi/lo/lo2)");
        hipFree(dA); hipFree(dB); hipFree(dC);
    }
    return 0;
}
