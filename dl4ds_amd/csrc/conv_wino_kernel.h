// dl4ds_amd -- Winograd F(2x2, 3x3) form of the MFMA-bound 3x3 convolutions (forward and, on the transposed filter, dgrad).
//
// The 3x3 layers of the residual backbone and of SubpixelConvolution (blocks.py:210-230, 433-454: 24..48 -> 24..192
// channels at 128^2 .. 256^2) run at 0.79 of the fp32 MFMA peak in conv_stream_ws_kernel: the direct form has nothing
// left to give.  Y = A^T [ (G g G^T) . (B^T d B) ] A spends 16 multiplications per 2x2 output tile and (cin, cout) pair
// instead of 36.  What made the transform lose inside the streaming kernel ("tried and dropped":
// half the MFMAs per streamed filter fragment -> vector-memory return path) is avoided by never moving the filter:
//   * a persistent workgroup (4 waves, one per SIMD, two workgroups per CU: 256 registers and 80 KB of LDS each) owns one
//     chunk of 16 NT output channels;
//     wave xi holds row xi of the TRANSFORMED filter U[xi][nu][cin][cout] (nu = 0..3, all cin of the pass, the
//     chunk's couts) as MFMA first operands in 16 KQ NT registers, computed once per launch from the 3x3 taps;
//   * per iteration the workgroup takes a "tile group" of 16 tiles (8 x 2 tiles = 16 x 4 output pixels):
//       A  every thread transforms one (tile, channel quad): 16 LDS reads of the raw 18 x 6 halo, 32 float4 add/sub,
//          16 LDS writes of V[xi nu][tile][cin]                                            -- barrier --
//       B  wave xi: 4 nu x 4 KQ k-steps x NT MFMAs (16x16x4: rows = couts, columns = the 16 tiles, K = cin; the pixel
//          operand is one ds_read_b128 per 4 k-steps, k-slot q owning channels 16 kq + 4 q + s), folds the four nu
//          products into the two columns of M A (R0 = M0 + M1 + M2, R1 = M1 - M2 - M3) and leaves them in LDS;
//          the halo of the NEXT tile group and the epilogue operands of this one were requested before the MFMAs
//          (zero-filling buffer loads, out-of-range offsets instead of branches) and are consumed after them
//                                                                                         -- barrier --
//       C  every thread finishes NT output quads: Y[i][j] = R[xi0][j] +- R[xi1][j] +- R[xi2][j] (rows of A^T), bias,
//          residual / ReLU / mask / accumulation, one 16-byte store; lanes walk a pixel row's channels -> whole
//          pixel segments per store instruction.
//   * input channels beyond 48 (the 192 -> 48 dgrad of conv2x) run as passes of 48 that accumulate into the output
//     (raw sums; the epilogue proper in the last pass).
// Issued multiply-adds: 16 per tile and (cin, cout) = 4 per output pixel instead of 9.  Numerics: fp32 throughout; the
// transforms add a few 1e-7 relative to the direct form (tests/test_gpu_ops.py compares both with the oracle).
// DL4DS_NO_WINOGRAD=1 restores the direct kernels everywhere.
// (Phase A as written above is the round-3 kernel, which staged V through LDS; the kernel in this header -- conv_wino2_kernel, "Second form" below --
// computes the input transform in the registers of the MFMA waves and has no phase A.  The round-3 kernel was an experiments-only alternative
// since round 4 and has been removed: git history, DESIGN_HISTORY.md.)
// This header holds the kernel; conv_wino_<KQ><NT>.hip instantiate one shape each (all epilogue forms), conv_wino.hip
// decides eligibility and passes.
#pragma once
#include "ops.h"
#include "prof.h"
#include "conv_kernels.h"
#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

struct WinoParams {
    ConvParams c;
    int cin0;               // first input channel of this pass: channels [cin0, cin0 + 16 KQ)
    const float* u;         // this pass's transformed filter in fragment order (wino_filter_kernel): [chunk][xi][f / 4][lane][4]
    int first;              // first pass over the input channels: the bias is added
    int nchunk;             // cout chunks of 16 NT
    int ntg, per_xcd;       // tile groups, and how many of them each XCD walks (a contiguous range)
    int tgx, tgy;           // tile groups per image row / column
    unsigned m_tgx, m_tgy;
#ifdef WINO_TRACE
    unsigned long long* trace;   // diagnostics build: [workgroup][wave][8] shader cycles per phase + iterations
#endif
};

// epilogue forms (compile-time: every runtime switch costs issue slots the MFMA waves of the other workgroup do not leave)
enum : int {
    WINO_OLDF = 1,          // the stored value is added BEFORE residual / ReLU / mask (passes > 0 over the input channels)
    WINO_ADD = 2,           // residual
    WINO_MASK = 4,          // result zeroed where mask <= 0 (ReLU backward)
    WINO_OLDA = 8,          // the stored value is added AFTER them (gradient accumulation)
};
inline bool wino_epi_built(int e) { return e == 0 || e == 1 || e == 2 || e == 3 || e == 4 || e == 5 || e == 8 || e == 12; }

// one entry point per shape (conv_wino_<KQ><NT>.hip)
void launch_wino_22(hipStream_t s, WinoParams& wp, int SX, int epi);
void launch_wino_23(hipStream_t s, WinoParams& wp, int SX, int epi);
void launch_wino_32(hipStream_t s, WinoParams& wp, int SX, int epi);
void launch_wino_33(hipStream_t s, WinoParams& wp, int SX, int epi);

namespace wino {

typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// a - b on float4 as two v_pk_add_f32 with negated second operand (the compiler emits four v_sub_f32)
__device__ __forceinline__ f32x4 sub4(const f32x4 a, const f32x4 b) {
    f32x2 lo, hi;
    const f32x2 alo = __builtin_shufflevector(a, a, 0, 1), ahi = __builtin_shufflevector(a, a, 2, 3);
    const f32x2 blo = __builtin_shufflevector(b, b, 0, 1), bhi = __builtin_shufflevector(b, b, 2, 3);
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(lo) : "v"(alo), "v"(blo));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(hi) : "v"(ahi), "v"(bhi));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}

// Packed arithmetic the COMPILER schedules (second forms of the kernels).  An inline-asm instruction is opaque to hipcc's hazard
// recogniser: placed next to MFMAs it may read an accumulator before the MFMA has written it, or overwrite a register an
// in-flight MFMA still reads (cdna_hip_programming.md section 5.7, item 2; the <2,2> weight-gradient shape failed that way, the
// <3,3> shape passed by luck of its register allocation).  <2 x float> fma / add select v_pk_fma_f32 / v_pk_add_f32; a
// subtraction is written as fma(b, -1, a) with an OPAQUE -1 (wino::pk_consts) so that it cannot be turned back into four
// scalar v_sub_f32, which is what hipcc does with float4 subtractions.  WINO_SCALAR_VALU builds the plain forms for comparison.
struct PkConsts { f32x2 neg1; };
__device__ __forceinline__ PkConsts pk_consts() {
    float m = -1.f;
    asm volatile("" : "+v"(m));                                      // (no instruction: only hides the value from the optimiser)
    PkConsts c;
    c.neg1 = (f32x2){m, m};
    return c;
}
__device__ __forceinline__ f32x2 pk_add(const f32x2 a, const f32x2 b) { return a + b; }
__device__ __forceinline__ f32x2 pk_fma(const f32x2 b, const f32x2 s, const f32x2 c) { return __builtin_elementwise_fma(b, s, c); }     // c + s b
__device__ __forceinline__ f32x2 pk_sub(const PkConsts& k, const f32x2 a, const f32x2 b) { return __builtin_elementwise_fma(b, k.neg1, a); }
__device__ __forceinline__ f32x2 lo2(const f32x4 a) { return __builtin_shufflevector(a, a, 0, 1); }
__device__ __forceinline__ f32x2 hi2(const f32x4 a) { return __builtin_shufflevector(a, a, 2, 3); }
__device__ __forceinline__ f32x4 cat4(const f32x2 l, const f32x2 h) { return __builtin_shufflevector(l, h, 0, 1, 2, 3); }
__device__ __forceinline__ f32x4 add4(const f32x4 a, const f32x4 b) {
#ifdef WINO_SCALAR_VALU
    return a + b;
#else
    return cat4(pk_add(lo2(a), lo2(b)), pk_add(hi2(a), hi2(b)));
#endif
}
__device__ __forceinline__ f32x4 subp4(const PkConsts& k, const f32x4 a, const f32x4 b) {
#ifdef WINO_SCALAR_VALU
    return a - b;
#else
    return cat4(pk_sub(k, lo2(a), lo2(b)), pk_sub(k, hi2(a), hi2(b)));
#endif
}
__device__ __forceinline__ f32x4 fma4(const f32x4 b, const f32x2 s, const f32x4 c) {       // c + s * b, s = (s, s)
#ifdef WINO_SCALAR_VALU
    return c + s[0] * b;
#else
    return cat4(pk_fma(lo2(b), s, lo2(c)), pk_fma(hi2(b), s, hi2(c)));
#endif
}

__device__ __forceinline__ void view_strides(const TView& v, size_t& sy, size_t& sx) {
    const int r = v.d2s > 1 ? v.d2s : 1;
    sx = (size_t)r * v.ld;
    sy = (size_t)r * (size_t)(v.W * r) * v.ld;
}

// ---------------------------------------------------------------------------------------------------------------------
// Second form (round 4): the input transform happens IN THE REGISTERS OF THE MFMA WAVES.
//
// The MFMA's second operand wants, in lane (column l15, k-slot lq), V[xi][nu][tile(l15)][cin 16 kq + 4 lq + s] -- which is
// exactly what a thread computes when it transforms (tile, channel quad 4 kq + lq) of the raw halo for ONE xi: two of the
// patch's four rows (xi 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3), four columns -> T[0..3], then V0 = T0 - T2,
// V1 = T1 + T2, V2 = T2 - T1, V3 = T1 - T3.  So wave xi reads 8 float4 per 16 channels straight from the raw halo and
// owns the four results as MFMA operands: V never exists in LDS (the first form wrote 16 and read 12 ds_*_b128 per thread
// and tile group for it), phase A and its barrier are gone, and the 32 additions per 16 channels sit in the gaps between
// the wave's own MFMAs.  The raw halo is double-buffered (the next one is DMA'd while this one is read by the K loop) at a
// pixel pitch of 4 KQ + 1 sixteen-byte slots: with the tile order below the ds_read_b128 lane groups ({0-3, 12-15,
// 20-27}, ...) cover all 16 slot columns exactly once -- no bank conflicts (the first form's linear pitch: 0.32-0.44 of
// its LDS cycles).  The DMA still fills LDS linearly; WHICH (pixel, quad) a lane fetches is its own offset, so any layout
// in 16-byte units is reachable, pad slots = out-of-range offsets.
//   MFMA column l15 <-> tile (ty, tx) of the 2 x 8 tiles of a group: columns 0-3, 12-15 = row 0 (tx = 0-3, 4-7), columns
//   4-11 = row 1.
// ---------------------------------------------------------------------------------------------------------------------
template <int KQ, int NT>
struct Wino2Geom {
    static constexpr int CK = 16 * KQ, Q4 = 4 * KQ, CO = 16 * NT, NQ = 4 * NT;
    static constexpr int SP = Q4 + 1;                               // raw pixel pitch in 16-byte slots
    static constexpr int HW = 18, HH = 6, HPIX = HW * HH;
    static constexpr int NSLOT = HPIX * SP, NCH = (NSLOT + 63) / 64, RAWS = NCH * 64;     // slots per buffer (whole DMA pieces)
    static constexpr int SIT = (NCH + 3) / 4;                       // DMA pieces per wave
    static constexpr int PP = CO + 8;                               // pitch of the folded products
    static constexpr int P = 8 * 16 * PP;                           // floats
    static constexpr int TAB = (SIT + (SIT + 3) / 4) * 256;          // per-thread DMA offsets and packed halo coordinates (ints)
    static constexpr size_t LDS_BYTES = (size_t)(2 * RAWS * 4 + P + TAB) * 4;
    static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
};

__device__ __forceinline__ int wino2_col_of_tile(int ty, int tx) { return ty ? tx + 4 : (tx < 4 ? tx : tx + 8); }

template <int KQ, int NT, int EPI>
__global__ void __launch_bounds__(256, 2) conv_wino2_kernel(const WinoParams wp) {
    typedef Wino2Geom<KQ, NT> GM;
    const ConvParams& a = wp.c;
    constexpr int Q4 = GM::Q4, CO = GM::CO, NQ = GM::NQ, SP = GM::SP, PP = GM::PP;
    constexpr int HW = GM::HW, HH = GM::HH, HPIX = GM::HPIX, SIT = GM::SIT, NCH = GM::NCH;
    constexpr int OOB = (int)0xffffff00u;
    constexpr int RSRC3 = 0x00020000;
    constexpr bool OLDF = (EPI & WINO_OLDF) != 0, ADD = (EPI & WINO_ADD) != 0, MASK = (EPI & WINO_MASK) != 0, OLDA = (EPI & WINO_OLDA) != 0;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const raw = lds;                                         // two buffers of RAWS slots
    float* const Pb = lds + 2 * GM::RAWS * 4;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, lq = lane >> 4;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, SX = gridDim.x >> 3;
    const int nsub = SX / wp.nchunk;
    const int chunk = slot % wp.nchunk, sub = slot / wp.nchunk;
    const int n0 = chunk * CO;
    const int tg_lo = xcd * wp.per_xcd, tg_hi = min(wp.ntg, tg_lo + wp.per_xcd);
    int tg = tg_lo + sub;
    if (sub >= nsub || tg >= tg_hi) return;                         // (whole workgroup)

    // ---- staging: DMA piece ch = 4 u + wave fills slots [64 ch, 64 ch + 64); slot L = (pixel L / SP, quad L % SP)
    size_t isy, isx;
    view_strides(a.in, isy, isx);
    // every thread's offsets live in LDS (six registers the K loop has no room for): tab[u][tid], then the packed coordinates
    int* const tab = reinterpret_cast<int*>(Pb + GM::P) + tid;
    {
        unsigned hyx[(SIT + 3) / 4];                                // (hy | hx << 3) of every piece's slot, a byte each: only border groups look
#pragma unroll
        for (int u = 0; u < (SIT + 3) / 4; ++u) hyx[u] = 0;
#pragma unroll
        for (int u = 0; u < SIT; ++u) {
            const int L = (4 * u + wave) * 64 + lane;
            const int p = L / SP, q = L - p * SP;
            const int cq = wp.cin0 + 4 * q;
            const int hy = p / HW, hx = p - hy * HW;
            const bool ok = q < Q4 && p < HPIX && cq < a.Cin;
            tab[u * 256] = ok ? (int)((hy * isy + hx * isx + view_chan_off(a.in, cq)) * 4) : OOB;
            hyx[u >> 2] |= (unsigned)(ok ? (hy | (hx << 3)) : 0) << (8 * (u & 3));
        }
#pragma unroll
        for (int u = 0; u < (SIT + 3) / 4; ++u) tab[(SIT + u) * 256] = (int)hyx[u];
    }
    struct Item { int n, y0, x0; };
    auto decode = [&](int t) {
        const int q = fast_div(t, wp.m_tgx);
        const int bx = t - q * wp.tgx;
        const int n = fast_div(q, wp.m_tgy);
        const int by = q - n * wp.tgy;
        Item it;
        it.n = n; it.y0 = by * 4; it.x0 = bx * 16;
        return it;
    };
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    auto stage_issue = [&](const Item& it, int buf) __attribute__((always_inline)) {
        const int ylo = max(0, 1 - it.y0), yhi = min(HH, a.H + 1 - it.y0);
        const int xlo = max(0, 1 - it.x0), xhi = min(HW, a.W + 1 - it.x0);
        int so[SIT];
#pragma unroll
        for (int u = 0; u < SIT; ++u) so[u] = tab[u * 256];
        if (ylo | xlo | (yhi - HH) | (xhi - HW)) {                  // border tile groups (the empty asm keeps it a branch)
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = 0; u < SIT; ++u) {
                const unsigned w = (unsigned)tab[(SIT + (u >> 2)) * 256];
                const int b = (int)((w >> (8 * (u & 3))) & 255u);
                const int hy = b & 7, hx = b >> 3;
                so[u] = (hy >= ylo && hy < yhi && hx >= xlo && hx < xhi) ? so[u] : OOB;
            }
        }
        const long org = (long)((size_t)it.n * a.in.nstride) + (long)(it.y0 - 1) * (long)isy + (long)(it.x0 - 1) * (long)isx;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(a.in.p)) + org * 4, 0, 0x7fffff00, RSRC3);
#if defined(__HIP_DEVICE_COMPILE__)                                   // (the host pass has no LDS address space to cast to)
        float* const dst = raw + buf * (GM::RAWS * 4) + wave * 256;
#pragma unroll
        for (int u = 0; u < SIT; ++u)
            if (4 * u + wave < NCH)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + u * 1024), 16, so[u], 0, 0, 0);
#else
        (void)rs; (void)so; (void)buf;
#endif
    };

    // ---- K loop operands: wave xi reads rows (ra, rb) of every tile's 4 x 4 patch, T = d[ra] + sg d[rb]
    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int rb = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sg1 = wave == 1 ? 1.f : -1.f;
    const f32x2 sg = {sg1, sg1};
    const PkConsts pkc = pk_consts();
    const int k_ty = (l15 >= 4 && l15 < 12) ? 1 : 0;
    const int k_tx = k_ty ? l15 - 4 : (l15 < 4 ? l15 : l15 - 8);
    const int k_base = ((2 * k_ty * HW + 2 * k_tx) * SP + lq) * 4;                  // floats
    const int k_a = k_base + ra * HW * SP * 4, k_b = k_base + rb * HW * SP * 4;

    // ---- phase C: thread owns output quads e = tid + 256 u = (pixel e / NQ, quad e % NQ)
    constexpr int ND = NT;
    size_t osy, osx;
    view_strides(a.out, osy, osx);
    int dvo[ND], prd[ND];
    const int nq = min(NQ, max(0, (a.Cout - n0) >> 2));
#pragma unroll
    for (int u = 0; u < ND; ++u) {
        const int e = tid + 256 * u;
        const int pix = e / NQ, quad = e - pix * NQ;
        const int py = pix >> 4, px = pix & 15;
        const int col = wino2_col_of_tile(py >> 1, px >> 1), i = py & 1, j = px & 1;
        prd[u] = (((i * 2 + j) * 16 + col) * PP + 4 * quad) | (i << 30);     // (bit 30: lower row -> r0 - (r1 + r2))
        dvo[u] = quad >= nq ? OOB : (int)((py * osy + px * osx + view_chan_off(a.out, n0 + 4 * quad)) * 4);
    }
    const float floor_v = a.relu ? 0.f : -3.0e38f;
    const bool want_bias = wp.first && a.bias != nullptr;
    const int bias_max = max(a.Cout - 4, 0);

    // ---- the wave's row of the transformed filter (as in the first form)
    constexpr int F = 16 * KQ * NT;
    float U[4][4 * KQ][NT];
    {
        const f32x4* up = reinterpret_cast<const f32x4*>(wp.u) + ((size_t)(chunk * 4 + wave) * (F / 4)) * 64 + lane;
#pragma unroll
        for (int f4 = 0; f4 < F / 4; ++f4) {
            f32x4 v = up[f4 * 64];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f = 4 * f4 + j;
                float x = v[j];
                asm volatile("" : "+v"(x));
                U[f / (4 * KQ * NT)][(f / NT) % (4 * KQ)][f % NT] = x;
            }
        }
    }

#ifdef WINO_TRACE
    unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tr_t = clock64();
#define WT(slot_) do { const unsigned long long n_ = clock64(); tr[slot_] += n_ - tr_t; tr_t = n_; } while (0)
#else
#define WT(slot_)
#endif
    Item cur = decode(tg);
    stage_issue(cur, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float* const pwr = Pb + ((wave * 2) * 16 + l15) * PP + 4 * lq;
    int buf = 0;
    WT(0);

    Item prev = cur;
    for (;;) {
        const int ntg = tg + nsub;
        const bool has_next = ntg < tg_hi;
        Item nxt = cur;
        // ---- K loop: per 16 channels 8 raw reads -> T[0..3] -> V0 = T0 - T2, V1 = T1 + T2, V2 = T2 - T1, V3 = T1 - T3 -> 4 x 4 x NT
        //      MFMAs.  The next 16 channels' columns are read one per MFMA step and row-transformed in the step after: 24
        //      ds_read_b128 and ~72 vector instructions per 144 MFMAs, all of them between the wave's own MFMAs.
        f32x4 acc[4][NT];
#pragma unroll
        for (int nu = 0; nu < 4; ++nu)
#pragma unroll
            for (int cb = 0; cb < NT; ++cb) acc[nu][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            const float* const pa = raw + buf * (GM::RAWS * 4) + k_a;
            const float* const pb = raw + buf * (GM::RAWS * 4) + k_b;
            f32x4 V[4], Tn[4], Da, Db;
            auto loadc = [&](int kq, int c) __attribute__((always_inline)) {         // column c of both rows
                Da = *reinterpret_cast<const f32x4*>(pa + (c * SP + 4 * kq) * 4);
                Db = *reinterpret_cast<const f32x4*>(pb + (c * SP + 4 * kq) * 4);
            };
            auto rowt = [&](int c) __attribute__((always_inline)) {
#ifdef WINO_NO_A
                Tn[c] = Da;
#else
                Tn[c] = fma4(Db, sg, Da);
#endif
            };
            auto colt = [&]() __attribute__((always_inline)) {
#ifdef WINO_NO_A
                V[0] = Tn[0]; V[1] = Tn[1]; V[2] = Tn[2]; V[3] = Tn[3];
#else
                V[0] = subp4(pkc, Tn[0], Tn[2]);
                V[1] = add4(Tn[1], Tn[2]);
                V[2] = subp4(pkc, Tn[2], Tn[1]);
                V[3] = subp4(pkc, Tn[1], Tn[3]);
#endif
            };
            auto mfmas = [&](int kq, int nu) __attribute__((always_inline)) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                    for (int cb = 0; cb < NT; ++cb)
                        acc[nu][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[nu][4 * kq + s4][cb], V[nu][s4], acc[nu][cb], 0, 0, 0);
            };
#pragma unroll
            for (int c = 0; c < 4; ++c) { loadc(0, c); rowt(c); }
            colt();
#pragma unroll
            for (int kq = 0; kq + 1 < KQ; ++kq) {
                __builtin_amdgcn_sched_barrier(0);
                loadc(kq + 1, 0);
                mfmas(kq, 0);
                __builtin_amdgcn_sched_barrier(0);
                rowt(0); loadc(kq + 1, 1);
                mfmas(kq, 1);
                if (kq == 0 && has_next) {
                    // the next halo: every wave left the K loop that read the other buffer before the last barrier
                    nxt = decode(ntg);
#ifndef WINO_NO_LOAD
                    stage_issue(nxt, buf ^ 1);
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
                rowt(1); loadc(kq + 1, 2);
                mfmas(kq, 2);
                __builtin_amdgcn_sched_barrier(0);
                rowt(2); loadc(kq + 1, 3);
                mfmas(kq, 3);
                rowt(3); colt();
            }
            // the last 16 channels: nu = 1, 2 first, so that the fold R0 = M0 + (M1 + M2), R1 = -M3 + (M1 - M2) happens between the
            // MFMAs of nu = 0, 3 instead of behind the stream (there every vector instruction waits for an MFMA slot of the other
            // workgroup's wave: 834 cycles for ~30 instructions in the trace)
            __builtin_amdgcn_sched_barrier(0);
            mfmas(KQ - 1, 1);
            if (KQ == 1 && has_next) {
                nxt = decode(ntg);
#ifndef WINO_NO_LOAD
                stage_issue(nxt, buf ^ 1);
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
            mfmas(KQ - 1, 2);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(KQ - 1, 0);
#pragma unroll
            for (int cb = 0; cb < NT; ++cb) {
                const f32x4 s12 = add4(acc[1][cb], acc[2][cb]);
                acc[2][cb] = subp4(pkc, acc[1][cb], acc[2][cb]);
                acc[1][cb] = s12;
            }
            __builtin_amdgcn_sched_barrier(0);
            mfmas(KQ - 1, 3);
#pragma unroll
            for (int cb = 0; cb < NT; ++cb) acc[0][cb] = add4(acc[0][cb], acc[1][cb]);
            __builtin_amdgcn_sched_barrier(0);
        }
        WT(2);
        f32x4 R0[NT], R1[NT];
#pragma unroll
        for (int cb = 0; cb < NT; ++cb) {
            R0[cb] = acc[0][cb];
            R1[cb] = add4(acc[3][cb], acc[2][cb]);
        }
        if (want_bias && wave == 1) {
            // rows 0 and 1 of A^T both carry xi = 1 with coefficient +1: the bias added to R[1][j] reaches all four outputs
#pragma unroll
            for (int cb = 0; cb < NT; ++cb) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + min(n0 + 16 * cb + 4 * lq, bias_max));
                R0[cb] = add4(R0[cb], b4);
                R1[cb] = add4(R1[cb], b4);
            }
        }
#pragma unroll
        for (int cb = 0; cb < NT; ++cb) {
            *reinterpret_cast<f32x4*>(pwr + 16 * cb) = R0[cb];
            *reinterpret_cast<f32x4*>(pwr + 16 * PP + 16 * cb) = R1[cb];
        }
        WT(3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's pieces of the next halo have landed
        WT(4);
        __syncthreads();                                            // products complete, next halo staged
        WT(5);
        // ---- C: Y = A^T (M A), epilogue, store
#ifndef WINO_NO_C
        {
            const int ymax = min(4, a.H - cur.y0), xmax = min(16, a.W - cur.x0);
            int dv[ND];
#pragma unroll
            for (int u = 0; u < ND; ++u) dv[u] = dvo[u];
            if ((ymax - 4) | (xmax - 16)) {                          // ragged right / bottom edge
                asm volatile("" ::: "memory");
#pragma unroll
                for (int u = 0; u < ND; ++u) {
                    const int pix = (tid + 256 * u) / NQ;
                    dv[u] = ((pix >> 4) < ymax && (pix & 15) < xmax) ? dvo[u] : OOB;
                }
            }
            const size_t pb_ = cur.y0 * osy + cur.x0 * osx;
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<char*>(a.out.p) + ((size_t)cur.n * a.out.nstride + pb_) * 4, 0, 0x7fffff00, RSRC3);
            i32x4_t e_old[(OLDF || OLDA) ? ND : 1], e_add[ADD ? ND : 1], e_mask[MASK ? ND : 1];
            if (OLDF || OLDA) {
#pragma unroll
                for (int u = 0; u < ND; ++u) e_old[u] = __builtin_amdgcn_raw_buffer_load_b128(ro, dv[u], 0, 0);
            }
            if (ADD) {
                const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc(
                    reinterpret_cast<char*>(a.add.p) + ((size_t)cur.n * a.add.nstride + pb_) * 4, 0, 0x7fffff00, RSRC3);
#pragma unroll
                for (int u = 0; u < ND; ++u) e_add[u] = __builtin_amdgcn_raw_buffer_load_b128(ra_, dv[u], 0, 0);
            }
            if (MASK) {
                const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(
                    reinterpret_cast<char*>(a.mask.p) + ((size_t)cur.n * a.mask.nstride + pb_) * 4, 0, 0x7fffff00, RSRC3);
#pragma unroll
                for (int u = 0; u < ND; ++u) e_mask[u] = __builtin_amdgcn_raw_buffer_load_b128(rm, dv[u], 0, 0);
            }
#pragma unroll
            for (int u = 0; u < ND; ++u) {
                const float* p = Pb + (prd[u] & 0x3fffffff);
                const f32x4 r0 = *reinterpret_cast<const f32x4*>(p);
                const f32x4 r1 = *reinterpret_cast<const f32x4*>(p + 2 * 16 * PP);
                const f32x4 r2 = *reinterpret_cast<const f32x4*>(p + 4 * 16 * PP);
                // (bit 30 of prd: the tile's lower row, Y = r0 - (r1 + r2); the sign rides on a multiply-add instead of a select)
                const float sgn = __builtin_bit_cast(float, 0x3f800000 | ((prd[u] << 1) & 0x80000000));
                f32x4 r = fma4(add4(r1, r2), (f32x2){sgn, sgn}, r0);
                if (OLDF) r = add4(r, __builtin_bit_cast(f32x4, e_old[u]));
                if (ADD) r = add4(r, __builtin_bit_cast(f32x4, e_add[u]));
                r[0] = fmaxf(r[0], floor_v); r[1] = fmaxf(r[1], floor_v); r[2] = fmaxf(r[2], floor_v); r[3] = fmaxf(r[3], floor_v);
                if (MASK) {
                    const f32x4 m = __builtin_bit_cast(f32x4, e_mask[u]);
                    r[0] = m[0] > 0.f ? r[0] : 0.f; r[1] = m[1] > 0.f ? r[1] : 0.f;
                    r[2] = m[2] > 0.f ? r[2] : 0.f; r[3] = m[3] > 0.f ? r[3] : 0.f;
                }
                if (OLDA) r = add4(r, __builtin_bit_cast(f32x4, e_old[u]));
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, r), ro, dv[u], 0, 0);
            }
        }
#endif
        WT(6);
#ifdef WINO_TRACE
        tr[0] += 1ull << 48;
#endif
        if (!has_next) break;
        cur = nxt;
        tg = ntg;
        buf ^= 1;
        __syncthreads();                                            // the products are consumed: P may be written again
        WT(7);
    }
    (void)prev;
#ifdef WINO_TRACE
    if (wp.trace && lane == 0)
        for (int q = 0; q < 8; ++q) wp.trace[((size_t)blockIdx.x * 4 + wave) * 8 + q] = tr[q];
#endif
}

template <int KQ, int NT, int EPI>
void launch_one(hipStream_t s, WinoParams& wp, int SX) {
#ifdef WINO_TRACE
    static unsigned long long* trace_buf = nullptr;
    static int trace_n = 0;
    wp.trace = nullptr;
    if (trace_n < 3) {
        if (!trace_buf) HIP_CHECK(hipMalloc((void**)&trace_buf, (size_t)1024 * 32 * 8));
        HIP_CHECK(hipMemsetAsync(trace_buf, 0, (size_t)1024 * 32 * 8, s));
        wp.trace = trace_buf;
    }
#endif
    {
        typedef Wino2Geom<KQ, NT> G2;
        static std::once_flag once2;
        std::call_once(once2, [&]() {
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino2_kernel<KQ, NT, EPI>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)G2::LDS_BYTES));
        });
        DL4DS_LAUNCH((conv_wino2_kernel<KQ, NT, EPI>), dim3(8 * SX), dim3(256), G2::LDS_BYTES, s, wp);
        HIP_CHECK(hipGetLastError());
#ifdef WINO_TRACE
        if (wp.trace) {
            ++trace_n;
            HIP_CHECK(hipStreamSynchronize(s));
            std::vector<unsigned long long> h((size_t)8 * SX * 32);
            HIP_CHECK(hipMemcpy(h.data(), trace_buf, h.size() * 8, hipMemcpyDeviceToHost));
            static const char* nm[8] = {"prologue", "issue", "K", "writeP", "landed", "bar1", "C", "bar2"};
            for (int wv = 0; wv < 4; wv += 3) {
                double sum[8] = {0}, its = 0;
                int nwg = 0;
                for (int b = 0; b < 8 * SX; ++b) {
                    const unsigned long long* t = &h[((size_t)b * 4 + wv) * 8];
                    const double it = (double)(t[0] >> 48);
                    if (it == 0) continue;
                    ++nwg; its += it;
                    for (int q = 0; q < 8; ++q) sum[q] += (double)(q == 0 ? (t[0] & ((1ull << 48) - 1)) : t[q]);
                }
                if (!nwg) continue;
                fprintf(stderr, "wino2<%d,%d,%d> wave %d: %d workgroups, %.1f iterations each; cycles: prologue %.0f | per iteration", KQ, NT, EPI,
                        wv, nwg, its / nwg, sum[0] / nwg);
                double tot = 0;
                for (int q = 1; q < 8; ++q) { fprintf(stderr, " %s %.0f", nm[q], sum[q] / its); tot += sum[q] / its; }
                fprintf(stderr, " = %.0f\n", tot);
            }
        }
#endif
    }
}

template <int KQ, int NT>
void launch_shape(hipStream_t s, WinoParams& wp, int SX, int epi) {
    switch (epi) {
        case 0: launch_one<KQ, NT, 0>(s, wp, SX); break;
        case 1: launch_one<KQ, NT, 1>(s, wp, SX); break;
        case 2: launch_one<KQ, NT, 2>(s, wp, SX); break;
        case 3: launch_one<KQ, NT, 3>(s, wp, SX); break;
        case 4: launch_one<KQ, NT, 4>(s, wp, SX); break;
        case 5: launch_one<KQ, NT, 5>(s, wp, SX); break;
        case 8: launch_one<KQ, NT, 8>(s, wp, SX); break;
        case 12: launch_one<KQ, NT, 12>(s, wp, SX); break;
        default: throw Dl4dsError("conv_wino: epilogue form " + std::to_string(epi) + " is not built");
    }
}

}  // namespace wino
