// dl4ds_amd -- 1x1 convolutions whose channel counts are not multiples of four.
//
// Replaces (for those shapes) tf.keras.layers.Conv2D(kernel_size=1) forward and its Conv2DBackpropInput -- call sites
// dl4ds/models/blocks.py:299 (TransitionBlock: densenet transitions C -> C // 2, TransitionLast 26 -> 13 of the recurrent
// nets), blocks.py:322 (LocalizedConvBlock's TransitionBlock(2)) and the projected skips of residual blocks.
//
// These layers are HBM streaming (26 -> 13: 338 MACs per pixel against 156 bytes) but none of the float4-staged kernels can
// take them: a pixel is 104 or 52 bytes, so neither the loads nor the stores of a channel quad are aligned, and a quad
// store would spill into the next pixel.  Here a block stages 256 CONSECUTIVE pixels -- for a plain view that is one
// contiguous, 16-byte aligned range -- with float4 loads straight into an LDS tile of the same layout (no index arithmetic), the GEMM [256 px x Cin] x [Cin x Cout]
// runs on the matrix cores with the whole (zero-padded) filter in registers, the results go through a second LDS tile and
// leave as one contiguous float4 stream again, the epilogue (bias, residual add, ReLU, ReLU-backward mask, gradient
// accumulation) applied element-wise on the way out.  Every byte is read once and written once.
#include "ops.h"
#include "prof.h"
#include <algorithm>
#include <cstdlib>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int PT = 256;          // pixels per tile

struct PointParams {
    const float* x; float* y;
    const float* add; const float* mask;
    const float* w; const float* bias;
    size_t npx;
    int Cin, Cout, relu, accumulate;
    int ldx;                     // pixel pitch of x in floats (> Cin: x is a channel slice of a wider buffer, e.g. a Concatenate's)
    int SX;                      // floats in the x tile (with slack, multiple of four)
    unsigned m_cin, m_cout;      // magic dividers
};

// exact e / d for e < 2^24-ish products used here (e < 256 * 64): 32-bit magic
__device__ __forceinline__ int divm(int n, unsigned magic) { return magic ? (int)__umulhi((unsigned)n, magic) : n; }

template <int KT, int NT>        // KT: k-steps of four input channels (Cin <= 4 KT); NT: tiles of 16 output channels
__global__ void __launch_bounds__(256) conv_point_kernel(const PointParams a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sx = sm;                              // [PT][Cin] + 4 KT floats of slack
    float* sy = sm + a.SX;                       // [PT][Cout]: the memory layout again
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int Cin = a.Cin, Cout = a.Cout, SY = a.Cout, LDX = a.ldx;

    // filter fragments: first MFMA operand = W^T[cout = 16 t + l15][k = 4 s + lq], zero beyond Cin / Cout
    float wr[NT][KT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int s = 0; s < KT; ++s) {
            const int k = 4 * s + lq, co = 16 * t + l15;
            const bool ok = k < Cin && co < Cout;
            const float v = a.w[(size_t)(ok ? k : 0) * Cout + (ok ? co : 0)];
            wr[t][s] = ok ? v : 0.f;
        }
    // LDS tiles keep the memory layout (pitch = channel count): staging and store are straight float4 copies.  The last k-step
    // of a pixel reads up to three floats beyond its channels -- the next pixel's first channels, times ZERO filter entries; the
    // floats behind the last staged pixel are zeroed so that nothing non-finite can be picked up there
    if (tid < 4 * KT) sx[PT * LDX + tid] = 0.f;
    // the bias comes from LDS in the epilogue: a global load there would wait for the prefetched loads of the next tile first
    // (vector memory loads return in order), which would expose their whole latency once per tile
    float* sb = sy + PT * SY;
    if (tid < 64) sb[tid] = (a.bias && tid < Cout) ? a.bias[tid] : 0.f;

    const size_t ntiles = (a.npx + PT - 1) / PT;
    // the tile's float4s go through registers: all of a thread's loads are issued back to back, and the NEXT tile's loads are in
    // flight while this one is multiplied and stored (a full tile is 64 Cin float4s: at most KT per thread)
    constexpr int RG = KT + 2;       // a full tile is 64 * ldx float4s; the launcher keeps ldx <= 4 KT + 8
    float4 rg[RG];
    auto issue = [&](size_t tile) __attribute__((always_inline)) {
        const size_t p0 = tile * PT;
        const int np = (a.npx - p0 < (size_t)PT) ? (int)(a.npx - p0) : PT;
        const float4* src = reinterpret_cast<const float4*>(a.x + p0 * LDX);
        const int n4 = ((np - 1) * LDX + Cin) >> 2;        // (the last pixel ends with its own channels: nothing is read beyond the view)
#pragma unroll
        for (int u = 0; u < RG; ++u) {
            const int i = tid + u * 256;
            rg[u] = (i < n4) ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    if ((size_t)blockIdx.x < ntiles) issue(blockIdx.x);
    for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const size_t p0 = tile * PT;
        const int np = (a.npx - p0 < (size_t)PT) ? (int)(a.npx - p0) : PT;
        // ---- the staged registers -> LDS (same layout as memory); a ragged end element by element
        {
            const int n = (np - 1) * LDX + Cin, n4 = n >> 2;
#pragma unroll
            for (int u = 0; u < RG; ++u) {
                const int i = tid + u * 256;
                if (i < n4) reinterpret_cast<float4*>(sx)[i] = rg[u];
            }
            const float* src = a.x + p0 * LDX;
            for (int e = 4 * n4 + tid; e < n; e += 256) sx[e] = src[e];
            if (tid < 4 * KT) sx[n + tid] = 0.f;          // what the last pixel's last k-step reads beyond its channels
        }
        __syncthreads();
        if (tile + gridDim.x < ntiles) issue(tile + gridDim.x);
        __syncthreads();
        // ---- [64 pixels per wave] x [Cout]: four pixel groups of 16, KT k-steps, NT cout tiles
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* bp = sx + (wave * 64 + g * 16 + l15) * LDX + lq;
#pragma unroll
            for (int s = 0; s < KT; ++s) {
                const float b = (4 * s < Cin) ? bp[4 * s] : 0.f;                 // (k-steps entirely beyond Cin: uniform skip)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[t][s], b, acc[t], 0, 0, 0);   // D[cout][pixel]
            }
            float* yp = sy + (wave * 64 + g * 16 + l15) * SY;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = 16 * t + 4 * lq + r;
                    if (co < Cout) yp[co] = acc[t][r];
                }
        }
        __syncthreads();
        // ---- epilogue + contiguous store of np * Cout floats
        {
            const size_t o0 = p0 * Cout;
            const int n = np * Cout, n4 = n >> 2;
            auto finish = [&](float v, int co, float addv, float maskv, float prev) __attribute__((always_inline)) {
                v += sb[co];
                if (a.add) v += addv;
                if (a.relu) v = fmaxf(v, 0.f);
                if (a.mask) v = maskv > 0.f ? v : 0.f;
                if (a.accumulate) v += prev;
                return v;
            };
            int co0 = 4 * tid - divm(4 * tid, a.m_cout) * Cout;               // channel of this thread's first quad
            const int dco = 1024 - divm(1024, a.m_cout) * Cout;               // ... and its advance per pass of 256 quads
            for (int i = tid; i < n4; i += 256) {
                const float4 yv = reinterpret_cast<const float4*>(sy)[i];
                float4 ad = make_float4(0.f, 0.f, 0.f, 0.f), mk = ad, pv = ad;
                if (a.add) ad = reinterpret_cast<const float4*>(a.add + o0)[i];
                if (a.mask) mk = reinterpret_cast<const float4*>(a.mask + o0)[i];
                if (a.accumulate) pv = reinterpret_cast<const float4*>(a.y + o0)[i];
                const float y4[4] = {yv.x, yv.y, yv.z, yv.w};
                const float ad4[4] = {ad.x, ad.y, ad.z, ad.w}, mk4[4] = {mk.x, mk.y, mk.z, mk.w}, pv4[4] = {pv.x, pv.y, pv.z, pv.w};
                float o4[4];
                int co = co0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o4[j] = finish(y4[j], co, ad4[j], mk4[j], pv4[j]);
                    co = (co + 1 >= Cout) ? co + 1 - Cout : co + 1;
                }
                reinterpret_cast<float4*>(a.y + o0)[i] = make_float4(o4[0], o4[1], o4[2], o4[3]);
                co0 += dco;
                if (co0 >= Cout) co0 -= Cout;
            }
            for (int e = 4 * n4 + tid; e < n; e += 256) {
                const int co = e - divm(e, a.m_cout) * Cout;
                a.y[o0 + e] = finish(sy[e], co, a.add ? a.add[o0 + e] : 0.f, a.mask ? a.mask[o0 + e] : 0.f,
                                     a.accumulate ? a.y[o0 + e] : 0.f);
            }
        }
        // (the next tile's staging writes sx only; sy is rewritten after the next barrier)
    }
}

bool plain(const TView& v) {
    return v.p && v.d2s <= 1 && !v.sc && v.ld == v.C && v.nstride == (size_t)v.H * v.W * v.C && ((((uintptr_t)v.p) & 15) == 0);
}

template <int KT, int NT>
void launch_point(hipStream_t s, const PointParams& p, size_t lds) {
    auto kern = conv_point_kernel<KT, NT>;
    static bool once = false;
    if (!once) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        once = true;
    }
    const size_t ntiles = (p.npx + PT - 1) / PT;
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, (size_t)(150 * 1024) / lds));
    const int blocks = (int)std::min<size_t>(ntiles, (size_t)256 * per_cu);
    DL4DS_LAUNCH(kern, dim3(blocks), dim3(256), lds, s, p);
    HIP_CHECK(hipGetLastError());
}

template <int KT>
void dispatch_nt(hipStream_t s, const PointParams& p, size_t lds) {
    const int nt = (p.Cout + 15) / 16;
    switch (nt) {
        case 1: launch_point<KT, 1>(s, p, lds); break;
        case 2: launch_point<KT, 2>(s, p, lds); break;
        case 3: launch_point<KT, 3>(s, p, lds); break;
        default: launch_point<KT, 4>(s, p, lds); break;
    }
}

}  // namespace

// 1x1, <= 64 -> <= 64 channels, at least one side not a multiple of four, plain contiguous views.  Returns false when the
// layer is not eligible (the caller goes on to the general kernels).
bool conv2d_point_forward(hipStream_t s, const TView& in, const float* w, int KS, const TView& out, const ConvEpilogue& ep) {
    if (KS != 1 || exp_env("DL4DS_NO_POINT")) return false;
    static const bool take_aligned = exp_env("DL4DS_POINT_FIRST") != nullptr;      // (experiment: aligned 1x1 layers here instead of conv_stream<1,..>)
    if ((!take_aligned && (in.C & 3) == 0 && (out.C & 3) == 0) || in.C > 64 || out.C > 64 || ep.pool) return false;
    // the input may be a channel slice of a wider buffer (pixel pitch ld > C: a Concatenate's buffer): whole pixels of the wide
    // buffer are staged, the channels outside the slice meet zero filter entries
    const bool in_ok = in.p && in.d2s <= 1 && !in.sc && in.ld >= in.C && in.nstride == (size_t)in.H * in.W * in.ld &&
                       ((((uintptr_t)in.p) & 15) == 0) && in.ld <= 4 * (4 * ((((in.C + 3) / 4) + 3) / 4)) + 8;
    if (!in_ok || !plain(out) || (ep.add.p && (!plain(ep.add) || ep.add.C != out.C)) ||
        (ep.mask.p && (!plain(ep.mask) || ep.mask.C != out.C)))
        return false;
    PointParams p;
    p.x = in.p; p.y = out.p; p.add = ep.add.p; p.mask = ep.mask.p; p.w = w; p.bias = ep.bias;
    p.npx = (size_t)in.N * in.H * in.W;
    p.Cin = in.C; p.Cout = out.C; p.relu = ep.relu; p.accumulate = ep.accumulate;
    const int kt = (in.C + 3) / 4;
    p.ldx = in.ld;
    p.SX = (PT * in.ld + 16 * ((kt + 3) / 4) + 3) & ~3;          // the kernel's k-steps come in groups of four (template KT)
    p.m_cin = div_magic(in.C);
    p.m_cout = div_magic(out.C);
    const size_t lds = ((size_t)p.SX + (size_t)PT * (out.C | 1) + 64) * sizeof(float);
    const double px = (double)p.npx;
    ProfScope ps(s, "conv_point", 2.0 * px * in.C * out.C,
                 4.0 * px * (in.ld + out.C * (1 + (ep.add.p ? 1 : 0) + (ep.mask.p ? 1 : 0) + (ep.accumulate ? 1 : 0))));
    switch ((kt + 3) / 4) {          // k-steps in groups of four: 16 / 32 / 48 / 64 input channels
        case 1: dispatch_nt<4>(s, p, lds); break;
        case 2: dispatch_nt<8>(s, p, lds); break;
        case 3: dispatch_nt<12>(s, p, lds); break;
        default: dispatch_nt<16>(s, p, lds); break;
    }
    return true;
}
