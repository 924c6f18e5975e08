// dl4ds_amd -- Winograd F(2x2, 3x3) form of the MFMA-bound 3x3 convolutions (forward and, on the transposed filter, dgrad).
//
// The 3x3 layers of the residual backbone and of SubpixelConvolution (blocks.py:210-230, 433-454: 24..48 -> 24..192
// channels at 128^2 .. 256^2) run at 0.79 of the fp32 MFMA peak in conv_stream_ws_kernel: the direct form has nothing
// left to give.  Y = A^T [ (G g G^T) . (B^T d B) ] A spends 16 multiplications per 2x2 output tile and (cin, cout) pair
// instead of 36.  What made the transform lose inside the streaming kernel ("tried and dropped":
// half the MFMAs per streamed filter fragment -> vector-memory return path) is avoided by never moving the filter:
//   * a persistent workgroup (4 waves, one per SIMD, one workgroup per CU) owns one chunk of 16 NT output channels;
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
#include "ops.h"
#include "prof.h"
#include "conv_kernels.h"
#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

namespace {

typedef int i32x4_t __attribute__((ext_vector_type(4)));

struct WinoParams {
    ConvParams c;
    int cin0;               // first input channel of this pass: channels / filter rows [cin0, cin0 + 16 KQ)
    int first, last;        // pass flags: bias in the first pass, the epilogue proper in the last; passes > 0 add the stored sums
    int nchunk;             // cout chunks of 16 NT
    int ntg, per_xcd;       // tile groups, and how many of them each XCD walks (a contiguous range)
    int tgx, tgy;           // tile groups per image row / column
    unsigned m_tgx, m_tgy;
#ifdef WINO_TRACE
    unsigned long long* trace;   // diagnostics build: [workgroup][wave][8] shader cycles per phase + iterations
#endif
};

template <int KQ, int NT>
struct WinoGeom {
    static constexpr int CK = 16 * KQ, Q4 = 4 * KQ, CO = 16 * NT, NQ = 4 * NT;
    static constexpr int RP = CK + 4;               // raw halo pixel pitch (floats)
    static constexpr int VP = CK + 8;               // V tile pitch: 2 mod 4 sixteen-byte slots (profiles/pmc_lds_pitch_r03.txt)
    static constexpr int PP = CO + 8;               // pitch of the folded products
    static constexpr int HW = 18, HH = 6, HPIX = HW * HH;
    static constexpr int RAW = HPIX * RP, V = 16 * 16 * VP, P = 8 * 16 * PP;      // floats
    static constexpr size_t LDS_BYTES = (size_t)(RAW + V + P) * 4;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

__device__ __forceinline__ void wino_view_strides(const TView& v, size_t& sy, size_t& sx) {
    const int r = v.d2s > 1 ? v.d2s : 1;
    sx = (size_t)r * v.ld;
    sy = (size_t)r * (size_t)(v.W * r) * v.ld;
}

template <int KQ, int NT>
__global__ void __launch_bounds__(256, 1) conv_wino_kernel(const WinoParams wp) {
    typedef WinoGeom<KQ, NT> GM;
    const ConvParams& a = wp.c;
    constexpr int Q4 = GM::Q4, CO = GM::CO, NQ = GM::NQ, RP = GM::RP, VP = GM::VP, PP = GM::PP;
    constexpr int HW = GM::HW, HH = GM::HH, HPIX = GM::HPIX;
    constexpr int OOB = (int)0xffffff00u;
    constexpr int RSRC3 = 0x00020000;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const raw = lds;
    float* const Vb = lds + GM::RAW;
    float* const Pb = Vb + GM::V;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, SX = gridDim.x >> 3;
    const int nsub = SX / wp.nchunk;
    const int chunk = slot % wp.nchunk, sub = slot / wp.nchunk;
    const int n0 = chunk * CO;
    const int tg_lo = xcd * wp.per_xcd, tg_hi = min(wp.ntg, tg_lo + wp.per_xcd);
    int tg = tg_lo + sub;
    if (sub >= nsub || tg >= tg_hi) return;                         // (whole workgroup)

    // ---- the wave's row of the transformed filter, as MFMA first operands: lane (row l15, k-slot lq), k-step ks = 4 kq + s
    //      -> U[xi = wave][nu][cin = cin0 + 16 kq + 4 lq + s][cout = n0 + 16 cb + l15]
    float U[4][4 * KQ][NT];
    {
        const float c0 = wave == 0 ? 1.f : (wave == 3 ? 0.f : .5f);
        const float c1 = wave == 1 ? .5f : (wave == 2 ? -.5f : 0.f);
        const float c2 = wave == 3 ? 1.f : (wave == 0 ? 0.f : .5f);
        const size_t tap = (size_t)a.Cin * a.Cout;
#pragma unroll
        for (int ks = 0; ks < 4 * KQ; ++ks) {
            const int cin = wp.cin0 + 16 * (ks >> 2) + 4 * lq + (ks & 3);
#pragma unroll
            for (int cb = 0; cb < NT; ++cb) {
                const int co = n0 + 16 * cb + l15;
                const bool ok = cin < a.Cin && co < a.Cout;
                const float* p = a.w + (ok ? (size_t)cin * a.Cout + co : 0);
                float t[3];
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const float g0 = ok ? p[(0 * 3 + b) * tap] : 0.f;
                    const float g1 = ok ? p[(1 * 3 + b) * tap] : 0.f;
                    const float g2 = ok ? p[(2 * 3 + b) * tap] : 0.f;
                    t[b] = c0 * g0 + c1 * g1 + c2 * g2;
                }
                U[0][ks][cb] = t[0];
                U[1][ks][cb] = .5f * (t[0] + t[1] + t[2]);
                U[2][ks][cb] = .5f * (t[0] - t[1] + t[2]);
                U[3][ks][cb] = t[2];
            }
        }
    }

    // ---- staging of the raw halo: thread = (channel quad, pixel p0 + PPASS u); zero padding = out-of-range offsets
    constexpr int PPASS = 256 / Q4, SIT = (HPIX + PPASS - 1) / PPASS;
    const int squad = tid % Q4, sp0 = tid / Q4;
    const bool st_active = sp0 < PPASS;
    size_t isy, isx;
    wino_view_strides(a.in, isy, isx);
    const int cq = wp.cin0 + 4 * squad;
    const bool q_ok = st_active && cq < a.Cin;
    const size_t in_c = q_ok ? view_chan_off(a.in, cq) : 0;
    int soff[SIT], hyx[SIT];
    auto rel_of = [&](int hy, int hx) { return (int)((hy * isy + hx * isx + in_c) * 4); };
#pragma unroll
    for (int u = 0; u < SIT; ++u) {
        const int hp = sp0 + PPASS * u;
        const int hy = hp / HW, hx = hp - hy * HW;
        const bool live = q_ok && hp < HPIX;
        hyx[u] = live ? ((hy << 8) | hx) : 0x7f7f;
        soff[u] = live ? rel_of(hy, hx) : OOB;
    }
    int st_sig = (HH << 8) | HW;
    const int st_dst = sp0 * RP + squad * 4;
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
    i32x4_t sr[SIT];
    auto stage_issue = [&](const Item& it) __attribute__((always_inline)) {
        const int ylo = max(0, 1 - it.y0), yhi = min(HH, a.H + 1 - it.y0);
        const int xlo = max(0, 1 - it.x0), xhi = min(HW, a.W + 1 - it.x0);
        const int sig = (ylo << 24) | (xlo << 16) | (yhi << 8) | xhi;
        if (sig != st_sig) {
            st_sig = sig;
#pragma unroll
            for (int u = 0; u < SIT; ++u) {
                const int hy = hyx[u] >> 8, hx = hyx[u] & 0xff;
                soff[u] = (hy >= ylo && hy < yhi && hx >= xlo && hx < xhi) ? rel_of(hy, hx) : OOB;
            }
        }
        const long org = (long)((size_t)it.n * a.in.nstride) + (long)(it.y0 - 1) * (long)isy + (long)(it.x0 - 1) * (long)isx;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(a.in.p)) + org * 4, 0, 0x7fffff00, RSRC3);
#pragma unroll
        for (int u = 0; u < SIT; ++u) sr[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, soff[u], 0, 0);
    };
    auto stage_write = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < SIT; ++u)
            if (u + 1 < SIT ? st_active : st_last) *reinterpret_cast<i32x4_t*>(raw + st_dst + u * (PPASS * RP)) = sr[u];
    };

    // ---- phase A: thread = (tile, channel quad)
    const bool a_on = tid < 16 * Q4;
    const int a_t = tid / Q4, a_q = tid - a_t * Q4;
    const int a_rd = (((a_t >> 3) * 2) * HW + (a_t & 7) * 2) * RP + 4 * a_q;
    const int a_wr = a_t * VP + 4 * a_q;

    // ---- phase C: thread owns output quads e = tid + 256 u = (pixel e / NQ, quad e % NQ)
    constexpr int ND = NT;
    size_t osy, osx;
    wino_view_strides(a.out, osy, osx);
    int dvo[ND], prd[ND];
    float sgn[ND];
    f32x4 bq[ND];
    const int nq = min(NQ, max(0, (a.Cout - n0) >> 2));
    auto out_off = [&](int u, int ymax, int xmax) {
        const int e = tid + 256 * u;
        const int pix = e / NQ, quad = e - pix * NQ;
        const int py = pix >> 4, px = pix & 15;
        if (py >= ymax || px >= xmax || quad >= nq) return OOB;
        return (int)((py * osy + px * osx + view_chan_off(a.out, n0 + 4 * quad)) * 4);
    };
#pragma unroll
    for (int u = 0; u < ND; ++u) {
        const int e = tid + 256 * u;
        const int pix = e / NQ, quad = e - pix * NQ;
        const int py = pix >> 4, px = pix & 15;
        const int t = (py >> 1) * 8 + (px >> 1), i = py & 1, j = px & 1;
        prd[u] = ((i * 2 + j) * 16 + t) * PP + 4 * quad;
        sgn[u] = i ? -1.f : 1.f;
        dvo[u] = out_off(u, 4, 16);
        bq[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (wp.first && a.bias && quad < nq) bq[u] = *reinterpret_cast<const f32x4*>(a.bias + n0 + 4 * quad);
    }
    int dr_sig = (4 << 8) | 16;
    const bool want_old = !wp.first || a.accumulate;
    const bool want_add = wp.last && a.add.p != nullptr, want_mask = wp.last && a.mask.p != nullptr;
    i32x4_t e_old[ND], e_add[ND], e_mask[ND];
    __amdgpu_buffer_rsrc_t ro;
    auto epi_prefetch = [&](const Item& it) __attribute__((always_inline)) {
        const int ymax = min(4, a.H - it.y0), xmax = min(16, a.W - it.x0);
        const int sig = (ymax << 8) | xmax;
        if (sig != dr_sig) {
            dr_sig = sig;
#pragma unroll
            for (int u = 0; u < ND; ++u) dvo[u] = out_off(u, ymax, xmax);
        }
        const size_t pb = it.y0 * osy + it.x0 * osx;
        ro = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.out.p) + ((size_t)it.n * a.out.nstride + pb) * 4, 0, 0x7fffff00, RSRC3);
        if (want_old) {
#pragma unroll
            for (int u = 0; u < ND; ++u) e_old[u] = __builtin_amdgcn_raw_buffer_load_b128(ro, dvo[u], 0, 0);
        }
        if (want_add) {
            const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<char*>(a.add.p) + ((size_t)it.n * a.add.nstride + pb) * 4, 0, 0x7fffff00, RSRC3);
#pragma unroll
            for (int u = 0; u < ND; ++u) e_add[u] = __builtin_amdgcn_raw_buffer_load_b128(ra, dvo[u], 0, 0);
        }
        if (want_mask) {
            const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<char*>(a.mask.p) + ((size_t)it.n * a.mask.nstride + pb) * 4, 0, 0x7fffff00, RSRC3);
#pragma unroll
            for (int u = 0; u < ND; ++u) e_mask[u] = __builtin_amdgcn_raw_buffer_load_b128(rm, dvo[u], 0, 0);
        }
    };

#ifdef WINO_TRACE
    unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tr_t = clock64();
#define WT(slot_) do { const unsigned long long n_ = clock64(); tr[slot_] += n_ - tr_t; tr_t = n_; } while (0)
#else
#define WT(slot_)
#endif
    Item cur = decode(tg);
    stage_issue(cur);
    stage_write();
    __syncthreads();
    const float* const vrd = Vb + ((wave * 4) * 16 + l15) * VP + 4 * lq;
    float* const pwr = Pb + ((wave * 2) * 16 + l15) * PP + 4 * lq;
    WT(0);
    for (;;) {
        // ---- A: V = B^T d B
        if (a_on) {
            f32x4 T[4][4];
            {
                f32x4 d[4][4];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) d[r][c] = *reinterpret_cast<const f32x4*>(raw + a_rd + (r * HW + c) * RP);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    T[0][c] = d[0][c] - d[2][c];
                    T[1][c] = d[1][c] + d[2][c];
                    T[2][c] = d[2][c] - d[1][c];
                    T[3][c] = d[1][c] - d[3][c];
                }
            }
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) {
                float* dst = Vb + (xi * 4 * 16) * VP + a_wr;
                *reinterpret_cast<f32x4*>(dst) = T[xi][0] - T[xi][2];
                *reinterpret_cast<f32x4*>(dst + 16 * VP) = T[xi][1] + T[xi][2];
                *reinterpret_cast<f32x4*>(dst + 32 * VP) = T[xi][2] - T[xi][1];
                *reinterpret_cast<f32x4*>(dst + 48 * VP) = T[xi][1] - T[xi][3];
            }
        }
        WT(1);
        __syncthreads();                                            // V complete, raw consumed
        WT(2);
        const int ntg = tg + nsub;
        const bool has_next = ntg < tg_hi;
        Item nxt = cur;
        if (has_next) {
            nxt = decode(ntg);
            stage_issue(nxt);
        }
        epi_prefetch(cur);
        WT(3);
        // ---- B: wave xi, M[nu] = U[xi][nu]^T V[xi][nu] over cin; R0 = M0 + M1 + M2, R1 = M1 - M2 - M3
        f32x4 R0[NT], R1[NT];
        {
            f32x4 av[4][KQ];
#pragma unroll
            for (int nu = 0; nu < 4; ++nu)
#pragma unroll
                for (int kq = 0; kq < KQ; ++kq) av[nu][kq] = *reinterpret_cast<const f32x4*>(vrd + nu * 16 * VP + 16 * kq);
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) {
                f32x4 acc[NT];
#pragma unroll
                for (int cb = 0; cb < NT; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kq = 0; kq < KQ; ++kq)
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                        for (int cb = 0; cb < NT; ++cb)
                            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[nu][4 * kq + s4][cb], av[nu][kq][s4], acc[cb], 0, 0, 0);
#pragma unroll
                for (int cb = 0; cb < NT; ++cb) {
                    if (nu == 0) R0[cb] = acc[cb];
                    else if (nu == 1) { R0[cb] += acc[cb]; R1[cb] = acc[cb]; }
                    else if (nu == 2) { R0[cb] += acc[cb]; R1[cb] -= acc[cb]; }
                    else R1[cb] -= acc[cb];
                }
            }
        }
#pragma unroll
        for (int cb = 0; cb < NT; ++cb) {
            *reinterpret_cast<f32x4*>(pwr + 16 * cb) = R0[cb];
            *reinterpret_cast<f32x4*>(pwr + 16 * PP + 16 * cb) = R1[cb];
        }
        WT(4);
        if (has_next) stage_write();
        WT(5);
        __syncthreads();                                            // products complete, next halo staged
        WT(6);
        // ---- C: Y = A^T (M A), epilogue, store
#pragma unroll
        for (int u = 0; u < ND; ++u) {
            const float* p = Pb + prd[u];
            const f32x4 r0 = *reinterpret_cast<const f32x4*>(p);
            const f32x4 r1 = *reinterpret_cast<const f32x4*>(p + 2 * 16 * PP);
            const f32x4 r2 = *reinterpret_cast<const f32x4*>(p + 4 * 16 * PP);
            f32x4 v = r0 + sgn[u] * (r1 + r2) + bq[u];
            if (!wp.first) v += __builtin_bit_cast(f32x4, e_old[u]);
            if (want_add) v += __builtin_bit_cast(f32x4, e_add[u]);
            if (wp.last && a.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            if (want_mask) {
                const f32x4 m = __builtin_bit_cast(f32x4, e_mask[u]);
                v[0] = m[0] > 0.f ? v[0] : 0.f; v[1] = m[1] > 0.f ? v[1] : 0.f;
                v[2] = m[2] > 0.f ? v[2] : 0.f; v[3] = m[3] > 0.f ? v[3] : 0.f;
            }
            if (wp.first && a.accumulate) v += __builtin_bit_cast(f32x4, e_old[u]);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, v), ro, dvo[u], 0, 0);
        }
        WT(7);
#ifdef WINO_TRACE
        tr[0] += 1ull << 48;
#endif
        if (!has_next) break;
        cur = nxt;
        tg = ntg;
    }
#ifdef WINO_TRACE
    if (wp.trace && lane == 0)
        for (int q = 0; q < 8; ++q) wp.trace[((size_t)blockIdx.x * 4 + wave) * 8 + q] = tr[q];
#endif
}

int wino_cu_count() {
    static const int n = [] {
        int dev = 0, v = 0;
        HIP_CHECK(hipGetDevice(&dev));
        HIP_CHECK(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
        return v;
    }();
    return n;
}

template <int KQ, int NT>
void launch_wino(hipStream_t s, WinoParams& wp, int SX) {
    typedef WinoGeom<KQ, NT> GM;
    static std::once_flag once;
    std::call_once(once, [&]() {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_kernel<KQ, NT>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)GM::LDS_BYTES));
    });
#ifdef WINO_TRACE
    static unsigned long long* trace_buf = nullptr;
    static int trace_n = 0;
    wp.trace = nullptr;
    if (trace_n < 12) {
        if (!trace_buf) HIP_CHECK(hipMalloc((void**)&trace_buf, (size_t)1024 * 32 * 8));
        HIP_CHECK(hipMemsetAsync(trace_buf, 0, (size_t)1024 * 32 * 8, s));
        wp.trace = trace_buf;
    }
#endif
    hipLaunchKernelGGL((conv_wino_kernel<KQ, NT>), dim3(8 * SX), dim3(256), GM::LDS_BYTES, s, wp);
    HIP_CHECK(hipGetLastError());
#ifdef WINO_TRACE
    if (wp.trace) {
        ++trace_n;
        HIP_CHECK(hipStreamSynchronize(s));
        std::vector<unsigned long long> h((size_t)8 * SX * 32);
        HIP_CHECK(hipMemcpy(h.data(), trace_buf, h.size() * 8, hipMemcpyDeviceToHost));
        static const char* nm[8] = {"prologue", "A", "bar1", "issue", "B", "stage_wr", "bar2", "C"};
        for (int wv = 0; wv < 4; ++wv) {
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
            fprintf(stderr, "wino<%d,%d> wave %d: %d workgroups, %.1f iterations each; cycles: prologue %.0f | per iteration", KQ, NT, wv, nwg,
                    its / nwg, sum[0] / nwg);
            double tot = 0;
            for (int q = 1; q < 8; ++q) { fprintf(stderr, " %s %.0f", nm[q], sum[q] / its); tot += sum[q] / its; }
            fprintf(stderr, " = %.0f\n", tot);
        }
    }
#endif
}

}  // namespace

// 3x3, stride 1, SAME.  Returns false when the layer is not eligible (the caller falls through to the direct kernels).
bool conv2d_wino_forward(hipStream_t s, const TView& in, const float* w, const TView& out, const ConvEpilogue& ep) {
    const bool off = getenv("DL4DS_NO_WINOGRAD") != nullptr;
    const char* force = getenv("DL4DS_WINO_FORCE");          // (tests: small grids too; "<k>" = k workgroups per XCD and cout chunk)
    if (off) return false;
    if (in.sc || ep.pool) return false;
    if (!in.vec || !out.vec || (in.C & 3) || (out.C & 3) || (ep.add.p && !ep.add.vec) || (ep.mask.p && !ep.mask.vec)) return false;
    if ((((uintptr_t)ep.bias) & 15) != 0) return false;
    if (in.C < 24 || out.C < 24) return false;
    auto same_layout = [](const TView& u, const TView& v) { return u.ld == v.ld && u.d2s == v.d2s && u.W == v.W && u.cp == v.cp; };
    if (ep.add.p && !same_layout(ep.add, out)) return false;
    if (ep.mask.p && !same_layout(ep.mask, out)) return false;
    auto span = [](const TView& v) { const size_t r = std::max(v.d2s, 1); return (size_t)8 * v.W * r * r * v.ld * 4; };
    if (span(in) >= (1ull << 31) || span(out) >= (1ull << 31)) return false;
    // channel passes
    int KQ, cpass;
    if (in.C <= 32) { KQ = 2; cpass = 32; }
    else if (in.C <= 48) { KQ = 3; cpass = 48; }
    else if (in.C % 48 == 0) { KQ = 3; cpass = 48; }
    else if (in.C % 32 == 0) { KQ = 2; cpass = 32; }
    else return false;
    const int passes = cdiv(in.C, cpass);
    if (passes > 1 && ep.accumulate) return false;
    const int NT = out.C <= 32 ? 2 : 3;
    WinoParams wp;
    ConvParams& p = wp.c;
    p.in = in; p.out = out; p.add = ep.add; p.mask = ep.mask;
    p.w = w; p.bias = ep.bias;
    p.Cin = in.C; p.Cout = out.C; p.H = in.H; p.W = in.W;
    p.relu = ep.relu; p.accumulate = ep.accumulate;
    wp.nchunk = cdiv(out.C, 16 * NT);
    wp.tgx = cdiv(in.W, 16);
    wp.tgy = cdiv(in.H, 4);
    wp.m_tgx = div_magic(wp.tgx);
    wp.m_tgy = div_magic(wp.tgy);
    const long ntg = (long)wp.tgx * wp.tgy * in.N;
    if (ntg >= (1l << 20)) return false;
    wp.ntg = (int)ntg;
    wp.per_xcd = cdiv(wp.ntg, 8);
    const int SXmax = std::max(wino_cu_count() / 8, 1);
    if (wp.nchunk > SXmax) return false;
    int SX = (SXmax / wp.nchunk) * wp.nchunk;
    if (force && atoi(force) > 0) SX = std::min(SX, atoi(force) * wp.nchunk);
    if (!force && (long)wp.per_xcd * wp.nchunk < 4l * SX) return false;       // fewer than four tile groups per workgroup
    const double px = (double)in.N * in.H * in.W;
    // (issued work: 16 multiply-adds per 2x2 tile and (cin, cout) pair of the padded operands, plus the transforms' additions)
    const double issued = 2.0 * (px / 4.0) * 16.0 * (16.0 * KQ * passes) * (16.0 * NT * wp.nchunk) +
                          (px / 4.0) * (32.0 * in.C * wp.nchunk + 24.0 * out.C * passes);
    ProfScope ps(s, "conv_wino<" + std::to_string(KQ) + "," + std::to_string(NT) + ">", issued,
                 4.0 * (px * (in.C + out.C * (1 + (ep.add.p ? 1 : 0) + (ep.mask.p ? 1 : 0) + (ep.accumulate ? 1 : 0))) + 9.0 * in.C * out.C));
    for (int ps_ = 0; ps_ < passes; ++ps_) {
        wp.cin0 = ps_ * cpass;
        wp.first = ps_ == 0;
        wp.last = ps_ == passes - 1;
        if (KQ == 2) {
            if (NT == 2) launch_wino<2, 2>(s, wp, SX); else launch_wino<2, 3>(s, wp, SX);
        } else {
            if (NT == 2) launch_wino<3, 2>(s, wp, SX); else launch_wino<3, 3>(s, wp, SX);
        }
    }
    return true;
}
