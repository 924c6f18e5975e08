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

template <int KQ, int NT>
struct WinoGeom {
    static constexpr int CK = 16 * KQ, Q4 = 4 * KQ, CO = 16 * NT, NQ = 4 * NT;
    static constexpr int RP = CK;                   // raw halo pixel pitch (floats): linear, it is filled by buffer_load ... lds
    static constexpr int VP = CK + 8;               // V tile pitch: 2 mod 4 sixteen-byte slots (profiles/pmc_lds_pitch_r03.txt)
    static constexpr int PP = CO + 8;               // pitch of the folded products
    static constexpr int HW = 18, HH = 6, HPIX = HW * HH;
    static constexpr int RAW = HPIX * RP, V = 16 * 16 * VP, P = 8 * 16 * PP;      // floats; the folded products overlay V
    static_assert(P <= V, "the folded products overlay V");
    static constexpr size_t LDS_BYTES = (size_t)(RAW + V) * 4;
    static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
};

__device__ __forceinline__ void view_strides(const TView& v, size_t& sy, size_t& sx) {
    const int r = v.d2s > 1 ? v.d2s : 1;
    sx = (size_t)r * v.ld;
    sy = (size_t)r * (size_t)(v.W * r) * v.ld;
}

template <int KQ, int NT, int EPI>
__global__ void __launch_bounds__(256, 2) conv_wino_kernel(const WinoParams wp) {
    typedef WinoGeom<KQ, NT> GM;
    const ConvParams& a = wp.c;
    constexpr int Q4 = GM::Q4, CO = GM::CO, NQ = GM::NQ, RP = GM::RP, VP = GM::VP, PP = GM::PP;
    constexpr int HW = GM::HW, HH = GM::HH, HPIX = GM::HPIX;
    constexpr int OOB = (int)0xffffff00u;
    constexpr int RSRC3 = 0x00020000;
    constexpr bool OLDF = (EPI & WINO_OLDF) != 0, ADD = (EPI & WINO_ADD) != 0, MASK = (EPI & WINO_MASK) != 0, OLDA = (EPI & WINO_OLDA) != 0;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const raw = lds;
    float* const Vb = lds + GM::RAW;
    float* const Pb = Vb;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, SX = gridDim.x >> 3;
    const int nsub = SX / wp.nchunk;
    const int chunk = slot % wp.nchunk, sub = slot / wp.nchunk;
    const int n0 = chunk * CO;
    const int tg_lo = xcd * wp.per_xcd, tg_hi = min(wp.ntg, tg_lo + wp.per_xcd);
    int tg = tg_lo + sub;
    if (sub >= nsub || tg >= tg_hi) return;                         // (whole workgroup)

    // ---- staging of the raw halo: thread = (channel quad, pixel p0 + PPASS u); zero padding = out-of-range offsets
    constexpr int PPASS = 256 / Q4, SIT = (HPIX + PPASS - 1) / PPASS;
    const int squad = tid % Q4, sp0 = tid / Q4;
    const bool st_active = sp0 < PPASS;
    size_t isy, isx;
    view_strides(a.in, isy, isx);
    const int cq = wp.cin0 + 4 * squad;
    const bool q_ok = st_active && cq < a.Cin;
    const size_t in_c = q_ok ? view_chan_off(a.in, cq) : 0;
    int soff[SIT];
    auto rel_of = [&](int hy, int hx) { return (int)((hy * isy + hx * isx + in_c) * 4); };
#pragma unroll
    for (int u = 0; u < SIT; ++u) {
        const int hp = sp0 + PPASS * u;
        const int hy = hp / HW, hx = hp - hy * HW;
        soff[u] = (q_ok && hp < HPIX) ? rel_of(hy, hx) : OOB;
    }
    const bool st_last = st_active && sp0 + PPASS * (SIT - 1) < HPIX;
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
    // the halo goes straight into LDS (buffer_load ... lds: wave-uniform base + 16 bytes per lane = the thread order above);
    // nothing is held in registers while the MFMAs run
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
#ifdef WINO_REGSTAGE
    constexpr bool REGSTAGE = 16 * KQ * NT <= 96;                  // (experiment: halo through registers where there is room)
#else
    constexpr bool REGSTAGE = false;
#endif
    i32x4_t sreg[REGSTAGE ? SIT : 1];
    const int st_wave = __builtin_amdgcn_readfirstlane(wave) * 256;            // floats
    auto stage_issue = [&](const Item& it) __attribute__((always_inline)) {
        const int ylo = max(0, 1 - it.y0), yhi = min(HH, a.H + 1 - it.y0);
        const int xlo = max(0, 1 - it.x0), xhi = min(HW, a.W + 1 - it.x0);
        int so[SIT];
#pragma unroll
        for (int u = 0; u < SIT; ++u) so[u] = soff[u];
        if (ylo | xlo | (yhi - HH) | (xhi - HW)) {                  // border tile groups (the empty asm keeps it a branch)
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = 0; u < SIT; ++u) {
                const int hp = sp0 + PPASS * u;
                const int hy = hp / HW, hx = hp - hy * HW;
                so[u] = (hy >= ylo && hy < yhi && hx >= xlo && hx < xhi) ? soff[u] : OOB;
            }
        }
        const long org = (long)((size_t)it.n * a.in.nstride) + (long)(it.y0 - 1) * (long)isy + (long)(it.x0 - 1) * (long)isx;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(a.in.p)) + org * 4, 0, 0x7fffff00, RSRC3);
        if (REGSTAGE) {
#pragma unroll
            for (int u = 0; u < SIT; ++u) sreg[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, so[u], 0, 0);
            return;
        }
#if defined(__HIP_DEVICE_COMPILE__)                                   // (the host pass has no LDS address space to cast to)
#pragma unroll
        for (int u = 0; u < SIT; ++u)
            if (u + 1 < SIT ? st_active : st_last)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(raw + st_wave + u * (PPASS * Q4 * 4)), 16, so[u], 0, 0, 0);
#else
        (void)rs; (void)st_wave; (void)st_last; (void)so;
#endif
    };
    auto stage_landed = [&]() __attribute__((always_inline)) {
        if (REGSTAGE) {
#pragma unroll
            for (int u = 0; u < SIT; ++u)
                if (u + 1 < SIT ? st_active : st_last) *reinterpret_cast<i32x4_t*>(raw + tid * 4 + u * (PPASS * Q4 * 4)) = sreg[u];
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };

    // ---- phase A: thread = (tile, channel quad)
    const bool a_on = tid < 16 * Q4;
    const int a_t = tid / Q4, a_q = tid - a_t * Q4;
    const int a_rd = (((a_t >> 3) * 2) * HW + (a_t & 7) * 2) * RP + 4 * a_q;
    const int a_wr = a_t * VP + 4 * a_q;

    // ---- phase C: thread owns output quads e = tid + 256 u = (pixel e / NQ, quad e % NQ)
    constexpr int ND = NT;
    size_t osy, osx;
    view_strides(a.out, osy, osx);
    int dvo[ND], prd[ND];
    const int nq = min(NQ, max(0, (a.Cout - n0) >> 2));
    auto out_off = [&](int u) {
        const int e = tid + 256 * u;
        const int pix = e / NQ, quad = e - pix * NQ;
        const int py = pix >> 4, px = pix & 15;
        if (quad >= nq) return OOB;
        return (int)((py * osy + px * osx + view_chan_off(a.out, n0 + 4 * quad)) * 4);
    };
    // (256 % NQ == 4 % NQ and 16 NQ pixels per row pair: whether an element belongs to the upper or the lower row of its tile
    //  -- the sign of rows 1, 2 of A^T -- is bit 4 of its pixel index)
#pragma unroll
    for (int u = 0; u < ND; ++u) {
        const int e = tid + 256 * u;
        const int pix = e / NQ, quad = e - pix * NQ;
        const int py = pix >> 4, px = pix & 15;
        const int t = (py >> 1) * 8 + (px >> 1), i = py & 1, j = px & 1;
        prd[u] = (((i * 2 + j) * 16 + t) * PP + 4 * quad) | (i << 30);        // (bit 30: lower row -> r0 - (r1 + r2))
        dvo[u] = out_off(u);
    }
    const float floor_v = a.relu ? 0.f : -3.0e38f;
    const bool want_bias = wp.first && a.bias != nullptr;
    const int bias_max = max(a.Cout - 4, 0);                        // (couts beyond Cout are never stored: any finite value will do)

    // ---- the wave's row of the transformed filter, as MFMA first operands: lane (row l15, k-slot lq), k-step ks = 4 kq + s
    //      -> U[xi = wave][nu][cin = cin0 + 16 kq + 4 lq + s][cout = n0 + 16 cb + l15]; nu = 3 is stored NEGATED: its
    //      products continue the accumulation of R1 = M1 - M2 - M3.  wino_filter_kernel left them in exactly this order.
    constexpr int F = 16 * KQ * NT;
    float U[4][4 * KQ][NT];
    {
        const f32x4* up = reinterpret_cast<const f32x4*>(wp.u) + ((size_t)(chunk * 4 + wave) * (F / 4)) * 64 + lane;
#pragma unroll
        for (int f4 = 0; f4 < F / 4; ++f4) {
            f32x4 v = up[f4 * 64];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                constexpr int dummy = 0; (void)dummy;
                const int f = 4 * f4 + j;
                float x = v[j];
                asm volatile("" : "+v"(x));                          // (opaque: a negation is not re-derived inside the loop)
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
    stage_issue(cur);
    stage_landed();
    __syncthreads();
    const float* const vrd = Vb + ((wave * 4) * 16 + l15) * VP + 4 * lq;
    float* const pwr = Pb + ((wave * 2) * 16 + l15) * PP + 4 * lq;
    WT(0);
#ifndef WINO_PRIO
#define WINO_PRIO 1
#endif
    for (;;) {
        // Vector-instruction issue on a SIMD is arbitrated by priority, then age (MI355X_MICROARCH.md, two waves per SIMD): at equal
        // priority the transform phases of the younger workgroup's wave get ONE issue per MFMA of the older one's phase B (3 000 cycles
        // for phase A's ~100 instructions).  Raised priority outside phase B lets them through at their own issue cost instead.
        if (WINO_PRIO) __builtin_amdgcn_s_setprio(WINO_PRIO);
        // ---- A: V = B^T d B (rows 1 and 2 of the halo feed all four xi)
#ifndef WINO_NO_A
        if (a_on) {
            const float* rp = raw + a_rd;
            float* vp = Vb + a_wr;
            auto emit = [&](int xi, const f32x4 (&T)[4]) __attribute__((always_inline)) {
                float* dst = vp + (xi * 4 * 16) * VP;
                *reinterpret_cast<f32x4*>(dst) = sub4(T[0], T[2]);
                *reinterpret_cast<f32x4*>(dst + 16 * VP) = T[1] + T[2];
                *reinterpret_cast<f32x4*>(dst + 32 * VP) = sub4(T[2], T[1]);
                *reinterpret_cast<f32x4*>(dst + 48 * VP) = sub4(T[1], T[3]);
            };
            f32x4 d1[4], d2[4], T[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                d1[c] = *reinterpret_cast<const f32x4*>(rp + (1 * HW + c) * RP);
                d2[c] = *reinterpret_cast<const f32x4*>(rp + (2 * HW + c) * RP);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) T[c] = d1[c] + d2[c];
            emit(1, T);
#pragma unroll
            for (int c = 0; c < 4; ++c) T[c] = sub4(d2[c], d1[c]);
            emit(2, T);
#pragma unroll
            for (int c = 0; c < 4; ++c) T[c] = sub4(*reinterpret_cast<const f32x4*>(rp + c * RP), d2[c]);
            emit(0, T);
#pragma unroll
            for (int c = 0; c < 4; ++c) T[c] = sub4(d1[c], *reinterpret_cast<const f32x4*>(rp + (3 * HW + c) * RP));
            emit(3, T);
        }
#endif
        WT(1);
        __syncthreads();                                            // V complete, raw consumed
        WT(2);
        const int ntg = tg + nsub;
        const bool has_next = ntg < tg_hi;
        Item nxt = cur;
        if (has_next) {
            nxt = decode(ntg);
#ifndef WINO_NO_LOAD
            stage_issue(nxt);
#endif
        }
        WT(3);
        // ---- B: wave xi, M[nu] = U[xi][nu]^T V[xi][nu] over cin; R0 = M0 + M1 + M2, R1 = M1 - M2 - M3.  nu = 0 accumulates in
        //      R0 and nu = 3 (negated filter) in R1 directly; M1 and M2 are added / subtracted by the vector unit
        f32x4 R0[NT], R1[NT];
        if (WINO_PRIO) __builtin_amdgcn_s_setprio(0);
#ifdef WINO_NO_B
#pragma unroll
        for (int cb = 0; cb < NT; ++cb) { R0[cb] = (f32x4){U[0][0][cb], 0.f, 0.f, 0.f}; R1[cb] = R0[cb]; }
#else
        {
            // the pixel operands of nu + 1 replace those of nu quad by quad, right after their last use: the LDS latency is
            // hidden behind the remaining MFMAs of nu without a second register set (the budget is 256 with two workgroups per CU)
            f32x4 av[KQ];
            auto kloop = [&](int nu, f32x4 (&acc)[NT]) __attribute__((always_inline)) {
#pragma unroll
                for (int kq = 0; kq < KQ; ++kq) {
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                        for (int cb = 0; cb < NT; ++cb)
                            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[nu][4 * kq + s4][cb], av[kq][s4], acc[cb], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (nu < 3) av[kq] = *reinterpret_cast<const f32x4*>(vrd + (nu + 1) * 16 * VP + 16 * kq);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            f32x4 M[NT];
#pragma unroll
            for (int kq = 0; kq < KQ; ++kq) av[kq] = *reinterpret_cast<const f32x4*>(vrd + 16 * kq);
#pragma unroll
            for (int cb = 0; cb < NT; ++cb) R0[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            kloop(0, R0);
#pragma unroll
            for (int cb = 0; cb < NT; ++cb) R1[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            kloop(1, R1);
#pragma unroll
            for (int cb = 0; cb < NT; ++cb) M[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            kloop(2, M);
#pragma unroll
            for (int cb = 0; cb < NT; ++cb) {
                R0[cb] += R1[cb] + M[cb];
                R1[cb] = sub4(R1[cb], M[cb]);
            }
            kloop(3, R1);
        }
#endif
        WT(4);
        if (WINO_PRIO) __builtin_amdgcn_s_setprio(WINO_PRIO);
        __syncthreads();                                            // every wave has read its rows of V: the products take its place
        if (want_bias && wave == 1) {
            // rows 0 and 1 of A^T both carry xi = 1 with coefficient +1: the bias added to R[1][j] reaches all four outputs
#pragma unroll
            for (int cb = 0; cb < NT; ++cb) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + min(n0 + 16 * cb + 4 * lq, bias_max));
                R0[cb] += b4;
                R1[cb] += b4;
            }
        }
#pragma unroll
        for (int cb = 0; cb < NT; ++cb) {
            *reinterpret_cast<f32x4*>(pwr + 16 * cb) = R0[cb];
            *reinterpret_cast<f32x4*>(pwr + 16 * PP + 16 * cb) = R1[cb];
        }
        stage_landed();
        WT(5);
        __syncthreads();                                            // products complete, next halo staged
        WT(6);
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
            const size_t pb = cur.y0 * osy + cur.x0 * osx;
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<char*>(a.out.p) + ((size_t)cur.n * a.out.nstride + pb) * 4, 0, 0x7fffff00, RSRC3);
            i32x4_t e_old[(OLDF || OLDA) ? ND : 1], e_add[ADD ? ND : 1], e_mask[MASK ? ND : 1];
            if (OLDF || OLDA) {
#pragma unroll
                for (int u = 0; u < ND; ++u) e_old[u] = __builtin_amdgcn_raw_buffer_load_b128(ro, dv[u], 0, 0);
            }
            if (ADD) {
                const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
                    reinterpret_cast<char*>(a.add.p) + ((size_t)cur.n * a.add.nstride + pb) * 4, 0, 0x7fffff00, RSRC3);
#pragma unroll
                for (int u = 0; u < ND; ++u) e_add[u] = __builtin_amdgcn_raw_buffer_load_b128(ra, dv[u], 0, 0);
            }
            if (MASK) {
                const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(
                    reinterpret_cast<char*>(a.mask.p) + ((size_t)cur.n * a.mask.nstride + pb) * 4, 0, 0x7fffff00, RSRC3);
#pragma unroll
                for (int u = 0; u < ND; ++u) e_mask[u] = __builtin_amdgcn_raw_buffer_load_b128(rm, dv[u], 0, 0);
            }
            f32x4 v[ND];
#pragma unroll
            for (int u = 0; u < ND; ++u) {
                const float* p = Pb + (prd[u] & 0x3fffffff);
                const f32x4 r0 = *reinterpret_cast<const f32x4*>(p);
                const f32x4 r1 = *reinterpret_cast<const f32x4*>(p + 2 * 16 * PP);
                const f32x4 r2 = *reinterpret_cast<const f32x4*>(p + 4 * 16 * PP);
                const f32x4 s12 = r1 + r2;
                v[u] = (prd[u] >> 30) ? sub4(r0, s12) : r0 + s12;
            }
#pragma unroll
            for (int u = 0; u < ND; ++u) {
                f32x4 r = v[u];
                if (OLDF) r += __builtin_bit_cast(f32x4, e_old[u]);
                if (ADD) r += __builtin_bit_cast(f32x4, e_add[u]);
                r[0] = fmaxf(r[0], floor_v); r[1] = fmaxf(r[1], floor_v); r[2] = fmaxf(r[2], floor_v); r[3] = fmaxf(r[3], floor_v);
                if (MASK) {
                    const f32x4 m = __builtin_bit_cast(f32x4, e_mask[u]);
                    r[0] = m[0] > 0.f ? r[0] : 0.f; r[1] = m[1] > 0.f ? r[1] : 0.f;
                    r[2] = m[2] > 0.f ? r[2] : 0.f; r[3] = m[3] > 0.f ? r[3] : 0.f;
                }
                if (OLDA) r += __builtin_bit_cast(f32x4, e_old[u]);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, r), ro, dv[u], 0, 0);
            }
        }
#endif
        WT(7);
#ifdef WINO_TRACE
        tr[0] += 1ull << 48;
#endif
        if (!has_next) break;
        cur = nxt;
        tg = ntg;
        __syncthreads();                                            // the products are consumed: V may be written again
    }
#ifdef WINO_TRACE
    if (wp.trace && lane == 0)
        for (int q = 0; q < 8; ++q) wp.trace[((size_t)blockIdx.x * 4 + wave) * 8 + q] = tr[q];
#endif
}

template <int KQ, int NT, int EPI>
void launch_one(hipStream_t s, WinoParams& wp, int SX) {
    typedef WinoGeom<KQ, NT> GM;
    static std::once_flag once;
    std::call_once(once, [&]() {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_kernel<KQ, NT, EPI>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)GM::LDS_BYTES));
    });
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
    hipLaunchKernelGGL((conv_wino_kernel<KQ, NT, EPI>), dim3(8 * SX), dim3(256), GM::LDS_BYTES, s, wp);
    HIP_CHECK(hipGetLastError());
#ifdef WINO_TRACE
    if (wp.trace) {
        ++trace_n;
        HIP_CHECK(hipStreamSynchronize(s));
        std::vector<unsigned long long> h((size_t)8 * SX * 32);
        HIP_CHECK(hipMemcpy(h.data(), trace_buf, h.size() * 8, hipMemcpyDeviceToHost));
        static const char* nm[8] = {"prologue", "A", "bar1", "issue", "B", "barP+writeP+landed", "bar2", "C"};
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
            fprintf(stderr, "wino<%d,%d,%d> wave %d: %d workgroups, %.1f iterations each; cycles: prologue %.0f | per iteration", KQ, NT, EPI,
                    wv, nwg, its / nwg, sum[0] / nwg);
            double tot = 0;
            for (int q = 1; q < 8; ++q) { fprintf(stderr, " %s %.0f", nm[q], sum[q] / its); tot += sum[q] / its; }
            fprintf(stderr, " = %.0f\n", tot);
        }
    }
#endif
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
