// dl4ds_amd -- stride-1 SAME convolution on SMALL grids with MANY channels as a GEMM over flattened pixels.
//
// The deep levels of unet_pin (dl4ds/models/sp_preups.py:262-285: 256 channels at 8 x 8, the 9 x 9 stride-2 transposed
// convolutions of DeconvolutionBlock -- blocks.py:508-533 -- as 5 x 5 convolutions 256 -> 4 x 256 on 8 x 8, 256 -> 4 x 128 on 16 x 16)
// have a few thousand pixels per batch and thousands of (tap, channel) terms per output.  The tile kernels cut an image into
// 16 x 16-pixel tiles (an 8 x 8 image fills a quarter of one) and get their parallelism from the images alone: 33-67 TFLOP/s.
// Here the M dimension is the flat list of pixels p = (n, y, x) of the whole batch -- no tile padding -- and the K dimension
// (taps x input channels) is split over blockIdx.z:
//     slab[z][p][co] = sum over the (tap, 16-channel chunk) steps of split z of  x[p + tap][chunk] . w[tap][chunk][co]
//   * a workgroup (4 waves, 2 x 2) owns 128 pixels x 128 output channels; per step it stages the 128 x 16 pixel operand
//     (16-byte zero-filling buffer loads: the tap's shift and the image border are the lane's offset) and the 16 x 128 filter
//     slice (transposed on its way into LDS) for the NEXT step into registers, runs 64 MFMAs per wave (16x16x4, rows =
//     output channels, columns = pixels: a lane ends with four consecutive output channels of one pixel) out of the current
//     LDS buffers and writes the registers to the other buffers: one barrier per step;
//   * both LDS operands are [row][16 k] at a pitch of 24 floats (6 sixteen-byte slots: k-slots one slot apart, rows 2 mod 4
//     slots apart -- conflict-free for the lane groups of ds_read_b128, profiles/pmc_lds_pitch_r03.txt), one read feeds the
//     four k-steps of a 16-channel chunk;
//   * the partial sums go to plain slabs; conv.hip's splitk_combine_kernel adds them in a fixed order (deterministic) and
//     applies bias / residual / ReLU / mask / accumulation through the real output view (plain or depth_to_space).
// DL4DS_NO_GEMM=1 leaves these layers to the tile kernels.
#include "ops.h"
#include "prof.h"
#include "conv_kernels.h"
#include <algorithm>
#include <mutex>
#include <string>

namespace {

constexpr int BM = 128, BN = 128, KC = 16, PITCH = 24;
constexpr int LDS_FLOATS = 2 * (BM + BN) * PITCH;

struct GemmParams {
    TView in;
    const float* w;         // [KK][Cin][Cout]
    float* slabs;           // [S][M][Cout]
    int H, W, Cin, Cout, KS, M;
    int nchunk_c;           // Cin / 16
    int nk;                 // KK * nchunk_c steps in all
    int cps;                // steps per split
    unsigned m_hw, m_w, m_cc;
};

typedef int i32x4_t __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256, 2) conv_gemm_kernel(const GemmParams a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int OOB = (int)0x7ffffff0;
    constexpr int RSRC3 = 0x00020000;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
    const int wm = wave & 1, wn = wave >> 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN, z = blockIdx.z;
    const int R = a.KS >> 1;
    const int q_lo = z * a.cps, q_hi = min(a.nk, q_lo + a.cps);

    // ---- pixel operand: thread = (pixel m0 + (tid >> 2) + 64 u, channel quad tid & 3); the tap's shift is added per step
    size_t isy, isx;
    {
        const int r = a.in.d2s > 1 ? a.in.d2s : 1;
        isx = (size_t)r * a.in.ld;
        isy = (size_t)r * (size_t)(a.in.W * r) * a.in.ld;
    }
    int px_y[2], px_x[2], px_off[2];
    const int xq = tid & 3;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int p = m0 + (tid >> 2) + 64 * u;
        if (p < a.M) {
            const int n = fast_div(p, a.m_hw);
            const int rem = p - n * (a.H * a.W);
            const int y = fast_div(rem, a.m_w), x = rem - y * a.W;
            px_y[u] = y; px_x[u] = x;
            px_off[u] = (int)((n * a.in.nstride + y * isy + x * isx) * 4);
        } else {
            px_y[u] = -1000; px_x[u] = -1000; px_off[u] = 0;          // (never inside the image: every tap reads zeros)
        }
    }
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(a.in.p)), 0,
                                                                        0x7ffffff0, RSRC3);
    // ---- filter operand: thread = (k = tid & 15, cout quad (tid >> 4) + 16 u): 16 rows of 64-byte pieces per wave
    const int wk = tid & 15, wq = tid >> 4;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(a.w)), 0,
                                                                        0x7ffffff0, RSRC3);
    int w_col[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int co = n0 + 4 * (wq + 16 * u);
        w_col[u] = co + 3 < a.Cout ? co * 4 : OOB;                   // (Cout % 4 == 0: whole quads or nothing)
    }
    float* const Xl = lds;                                           // [2][BM][PITCH]
    float* const Wl = lds + 2 * BM * PITCH;                          // [2][BN][PITCH]

    i32x4_t xreg[2], wreg[2];
    auto issue = [&](int q) __attribute__((always_inline)) {
        const int tap = fast_div(q, a.m_cc);
        const int cc = q - tap * a.nchunk_c;
        const int ty = tap / a.KS, tx = tap - ty * a.KS;
        const int dy = ty - R, dx = tx - R;
        const int shift = (int)(((long)dy * (long)isy + (long)dx * (long)isx + (long)view_chan_off(a.in, cc * KC + 4 * xq)) * 4);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int yy = px_y[u] + dy, xx = px_x[u] + dx;
            const bool ok = yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
            xreg[u] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? px_off[u] + shift : OOB, 0, 0);
        }
        const int wrow = (int)((((size_t)tap * a.Cin + cc * KC + wk) * a.Cout) * 4);
#pragma unroll
        for (int u = 0; u < 2; ++u)
            wreg[u] = __builtin_amdgcn_raw_buffer_load_b128(rw, w_col[u] == OOB ? OOB : wrow + w_col[u], 0, 0);
    };
    auto commit = [&](int b) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            *reinterpret_cast<i32x4_t*>(Xl + (b * BM + (tid >> 2) + 64 * u) * PITCH + 4 * xq) = xreg[u];
            float* wd = Wl + (b * BN + 4 * (wq + 16 * u)) * PITCH + wk;
            const f32x4 v = __builtin_bit_cast(f32x4, wreg[u]);
            wd[0] = v[0]; wd[PITCH] = v[1]; wd[2 * PITCH] = v[2]; wd[3 * PITCH] = v[3];
        }
    };

    f32x4 acc[4][4];                                                 // [cout block][pixel block]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (q_lo < q_hi) {
        issue(q_lo);
        commit(0);
        __syncthreads();
        for (int q = q_lo; q < q_hi; ++q) {
            const int b = (q - q_lo) & 1;
            const bool more = q + 1 < q_hi;
            if (more) issue(q + 1);
            const float* xa = Xl + (b * BM + wm * 64 + l15) * PITCH + 4 * lq;
            const float* wa = Wl + (b * BN + wn * 64 + l15) * PITCH + 4 * lq;
            f32x4 xf[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                wf[i] = *reinterpret_cast<const f32x4*>(wa + i * 16 * PITCH);
                xf[i] = *reinterpret_cast<const f32x4*>(xa + i * 16 * PITCH);
            }
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i][s4], xf[j][s4], acc[i][j], 0, 0, 0);
            if (more) commit(b ^ 1);
            __syncthreads();
        }
    }
    // ---- partial sums: lane (column l15 = pixel, rows 4 lq + r = four consecutive output channels)
    float* const slab = a.slabs + (size_t)z * a.M * a.Cout;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = m0 + wm * 64 + 16 * j + l15;
        if (p >= a.M) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = n0 + wn * 64 + 16 * i + 4 * lq;
            if (co + 3 < a.Cout) *reinterpret_cast<f32x4*>(slab + (size_t)p * a.Cout + co) = acc[i][j];
        }
    }
}

int gemm_cu_count() {
    static const int n = [] {
        int dev = 0, v = 0;
        HIP_CHECK(hipGetDevice(&dev));
        HIP_CHECK(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
        return v;
    }();
    return n;
}

}  // namespace

// false = not eligible (the caller goes on to the tile kernels)
bool conv2d_gemm_forward(hipStream_t s, const TView& in, const float* w, int KS, const TView& out, const ConvEpilogue& ep, int max_hw) {
    static const bool off = exp_env("DL4DS_NO_GEMM") != nullptr;
    if (off) return false;
    if (!(KS == 1 || KS == 3 || KS == 5)) return false;
    if (in.sc || ep.pool) return false;
    if (in.H * in.W > max_hw) return false;                               // larger grids fill the tile kernels
    if ((in.C & 15) || (out.C & 3) || in.C < 64 || out.C < 64) return false;
    if (!in.vec || (((uintptr_t)w) & 15)) return false;
    if (in.d2s > 1 && (in.cp & 15)) return false;                         // a 16-channel chunk stays inside one depth_to_space group
    const long M = (long)in.N * in.H * in.W;
    if (M >= (1l << 20) || M < 256) return false;
    // 32-bit byte offsets into the input and the filter
    const size_t r = std::max(in.d2s, 1);
    if ((size_t)in.N * in.nstride * 4 + (size_t)8 * in.W * r * r * in.ld * 4 >= (1ull << 31)) return false;
    if ((size_t)KS * KS * in.C * out.C * 4 >= (1ull << 31)) return false;
    GemmParams p;
    p.in = in; p.w = w;
    p.H = in.H; p.W = in.W; p.Cin = in.C; p.Cout = out.C; p.KS = KS; p.M = (int)M;
    p.nchunk_c = in.C / KC;
    p.nk = KS * KS * p.nchunk_c;
    p.m_hw = div_magic(in.H * in.W); p.m_w = div_magic(in.W); p.m_cc = div_magic(p.nchunk_c);
    const int gm = cdiv((int)M, BM), gn = cdiv(out.C, BN);
    // split K until every CU has about two workgroups, at least eight steps per split
    int S = std::max(1, std::min(p.nk / 8, cdiv(2 * gemm_cu_count(), gm * gn)));
    p.cps = cdiv(p.nk, S);
    S = cdiv(p.nk, p.cps);
    const size_t slab = (size_t)M * out.C;
    p.slabs = conv_splitk_scratch(s, slab * S);
    static std::once_flag once;
    std::call_once(once, [&]() {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)(LDS_FLOATS * sizeof(float))));
    });
    const double px = (double)M;
    ProfScope ps(s, "conv_gemm<" + std::to_string(KS) + ">", 2.0 * px * KS * KS * in.C * out.C,
                 4.0 * (px * (in.C + out.C * (1 + (ep.add.p ? 1 : 0) + (ep.mask.p ? 1 : 0) + (ep.accumulate ? 1 : 0))) + (double)KS * KS * in.C * out.C));
    DL4DS_LAUNCH(conv_gemm_kernel, dim3(gm, gn, S), dim3(256), LDS_FLOATS * sizeof(float), s, p);
    HIP_CHECK(hipGetLastError());
    ConvParams c;
    c.in = in; c.out = out; c.add = ep.add; c.mask = ep.mask;
    c.w = w; c.bias = ep.bias;
    c.Cin = in.C; c.Cout = out.C; c.H = in.H; c.W = in.W;
    c.relu = ep.relu; c.accumulate = ep.accumulate;
    c.wvec = 0; c.CK = 0; c.TPS = 1; c.tiles_x = c.tiles_y = 0;
    conv_splitk_combine(s, p.slabs, S, slab, c, in.N);
    return true;
}
