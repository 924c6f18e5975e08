// dl4ds_amd -- Winograd F(2x2, 3x3) weight gradient of the MFMA-bound 3x3 layers.
//
// dW = G^T [ sum over tiles (B^T d B) . (A dY A^T) ] G: 16 multiply-adds per 2x2 output tile and (cin, cout) pair instead of
// the 36 of the direct form (conv_wgrad_rows_ws_kernel runs those at 0.7-0.85 of the fp32 MFMA peak; blocks.py:49-61, 210-230,
// 414-416 in backward).  The shape mirrors conv_wino_kernel (conv_wino_kernel.h) with the roles turned round: what stays in
// registers is the ACCUMULATOR dU[xi][nu][cin][cout] -- wave xi of a persistent workgroup owns row xi for one (chunk of 16 KQ
// input channels, chunk of 16 NT output channels) pair, 16 KQ NT registers.  The kernel (conv_wino_wgrad2_kernel, below) forms both
// MFMA operands -- V = B^T d B from the raw 18 x 6 input halo, dM = A dY A^T from the raw 16 x 4 block of the output gradient, the
// two negative rows of A stored positive -- in the registers of the MFMA waves; the raw blocks are all LDS holds (72 KB, two workgroups
// per CU).  (The round-3 form that staged V and dM through LDS -- 106 KB, one workgroup per CU, two transform phases with the matrix
// side idle -- was an experiments-only alternative since round 4 and is gone: git history, DESIGN_HISTORY.md.)
// Every workgroup leaves its share in a slab -- the column half of G^T . G already applied, 12 planes -- wino_slab_sum_kernel adds the
// slabs of each (cin chunk, cout chunk) pair in a fixed order (bitwise reproducible) and wino_wgrad_finish_kernel applies the row half
// and the remaining sign, and writes / accumulates dW and db.  DL4DS_NO_WINOGRAD=1 or DL4DS_NO_WINOGRAD_WGRAD=1: direct kernels.
#include "conv_wino_kernel.h"

namespace {

using wino::sub4;
typedef int i32x4_t __attribute__((ext_vector_type(4)));

struct WinoWgradParams {
    TView x, dy;
    float* slab;            // [workgroup k][pair][FR + CO]: u[xi][b] in fragment order (12 planes), then the bias-gradient partials
    int Cin, Cout, H, W;
    int ncin, ncout;        // chunks of 16 KQ input / 16 NT output channels
    int ntg, per_xcd, tgx, tgy;
    unsigned m_tgx, m_tgy;
};

// ---------------------------------------------------------------------------------------------------------------------
// Second form (round 4): BOTH transforms in the registers of the MFMA waves, nothing but the two raw blocks in LDS.
//
// The MFMA operands of dU[xi][nu] += V[xi][nu]^T dM[xi][nu] are, in lane (l15, k-slot lq) and k-step s,
//     A = V [xi][nu][tile(s, lq)][cin  16 i + l15]        B = dM[xi][nu][tile(s, lq)][cout 16 j + l15]
// i.e. ONE channel of one tile each -- scalars.  A lane therefore reads the raw values of its own (tile, channel) pairs (8 of the
// input patch: rows ra, rb of wave xi, four columns; 4 of the output gradient's 2 x 2 pixels).  Channel blocks are taken TWO AT A
// TIME -- channels c and c + 16 of a pixel come back from one ds_read2_b32 as a register pair -- and transformed with packed
// instructions on those pairs (an odd third block with plain ones):
//     T_c = a_c + sg b_c (c = 0..3);  V0 = T0 - T2, V1 = T1 + T2, V2 = T2 - T1, V3 = T1 - T3
//     r_c = y_first,c + cy y_1,c (c = 0, 1; cy = 0, 1, -1, 0 for xi = 0..3, xi = 3 reads row 1 as its first row);
//     dM0 = r0, dM1 = r0 + r1, dM2 = r0 - r1, dM3 = r1
// all of it arithmetic the compiler schedules (no inline asm beside MFMAs: conv_wino_kernel.h, pk_consts).
// 8 + 4 packed instructions and 8 + 4 LDS reads per (tile, channel pair) feed 4 x 2 x (NT | KQ) MFMAs.  What the first form paid for V
// and dM -- 32 ds_write_b128 and 48 ds_read2 per thread and tile group, two transform phases with every wave idle on the matrix
// side, 106 KB of LDS that kept a CU to ONE workgroup -- is gone: 72 KB per workgroup (both raw blocks double-buffered + the
// DMA offset tables), two workgroups per CU, one barrier per tile group.  The k-slots of a k-step take the tiles tx = 2 (s & 1)
// + {0, 4, 1, 5}[lq] of tile row s >> 1: tiles four apart sit 32 banks (mod 64) apart at the raw pitch of 4 K + 1 sixteen-byte
// slots, so the 32-lane halves of every 8-byte read are conflict-free.  Slab layout and slab sum are unchanged; the closing
// transform un-permutes the channels of paired blocks.
// ---------------------------------------------------------------------------------------------------------------------
template <int KQ, int NT>
struct Wg2Geom {
    static constexpr int CK = 16 * KQ, Q4 = 4 * KQ, CO = 16 * NT, NQ = 4 * NT;
    static constexpr int SPX = Q4 + 1, SPY = NQ + 1;                 // raw pixel pitches in 16-byte slots
    static constexpr int HW = 18, HH = 6, HPIX = HW * HH, YPIX = 64;
    static constexpr int NCHX = (HPIX * SPX + 63) / 64, NCHY = (YPIX * SPY + 63) / 64;      // DMA pieces (64 slots each)
    static constexpr int RAWX = NCHX * 64, RAWY = NCHY * 64;        // slots per buffer
    static constexpr int SITX = (NCHX + 3) / 4, SITY = (NCHY + 3) / 4;
    static constexpr int TAB = (SITX + SITY) * 256;                 // ints
    static constexpr size_t LDS_BYTES = (size_t)(2 * (RAWX + RAWY) * 4 + TAB) * 4;
    static constexpr int FR = 4 * (3 * KQ * NT) * 64 * 4;
    static constexpr int ST = FR + CO;
    static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
};

typedef float f32x2_t __attribute__((ext_vector_type(2)));

template <int KQ, int NT>
__global__ void __launch_bounds__(256, 2) conv_wino_wgrad2_kernel(const WinoWgradParams wp) {
    typedef Wg2Geom<KQ, NT> GM;
    constexpr int CK = GM::CK, Q4 = GM::Q4, CO = GM::CO, NQ = GM::NQ, SPX = GM::SPX, SPY = GM::SPY;
    constexpr int HW = GM::HW, HH = GM::HH, HPIX = GM::HPIX, SITX = GM::SITX, SITY = GM::SITY, NCHX = GM::NCHX, NCHY = GM::NCHY;
    constexpr int OOB = (int)0xffffff00u;
    constexpr int RSRC3 = 0x00020000;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const rawx = lds;                                        // two buffers of RAWX slots
    float* const rawy = lds + 2 * GM::RAWX * 4;                     // two buffers of RAWY slots

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, lq = lane >> 4;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, SX = gridDim.x >> 3;
    const int npair = wp.ncin * wp.ncout, nsub = SX / npair;
    const int pc = slot % npair, sub = slot / npair;
    if (sub >= nsub) return;
    const int cin0 = (pc / wp.ncout) * CK, n0 = (pc % wp.ncout) * CO;
    const int tg_lo = xcd * wp.per_xcd, tg_hi = min(wp.ntg, tg_lo + wp.per_xcd);
    int tg = tg_lo + sub;
    float* const myslab = wp.slab + ((size_t)(xcd * nsub + sub) * npair + pc) * GM::ST;

    // ---- staging: DMA piece ch = 4 u + wave fills slots [64 ch, 64 ch + 64) of a raw buffer; slot L = (pixel L / SP, quad L % SP).
    //      Every thread's offsets live in LDS (tab[u][tid]); the packed (y, x) of its slots stay in three registers.
    size_t isy, isx, ysy, ysx;
    wino::view_strides(wp.x, isy, isx);
    wino::view_strides(wp.dy, ysy, ysx);
    int* const tab = reinterpret_cast<int*>(rawy + 2 * GM::RAWY * 4) + tid;
    unsigned hyx[(SITX + 3) / 4], hyy[(SITY + 3) / 4];
    {
#pragma unroll
        for (int u = 0; u < (SITX + 3) / 4; ++u) hyx[u] = 0;
#pragma unroll
        for (int u = 0; u < (SITY + 3) / 4; ++u) hyy[u] = 0;
        const int nq = min(NQ, max(0, (wp.Cout - n0) >> 2));
#pragma unroll
        for (int u = 0; u < SITX; ++u) {
            const int L = (4 * u + wave) * 64 + lane;
            const int p = L / SPX, q = L - p * SPX;
            const int cq = cin0 + 4 * q;
            const int hy = p / HW, hx = p - hy * HW;
            const bool ok = q < Q4 && p < HPIX && cq < wp.Cin;
            tab[u * 256] = ok ? (int)((hy * isy + hx * isx + view_chan_off(wp.x, cq)) * 4) : OOB;
            hyx[u >> 2] |= (unsigned)(ok ? (hy | (hx << 3)) : 0) << (8 * (u & 3));
        }
#pragma unroll
        for (int u = 0; u < SITY; ++u) {
            const int L = (4 * u + wave) * 64 + lane;
            const int p = L / SPY, q = L - p * SPY;
            const int py = p >> 4, px = p & 15;
            const bool ok = q < nq && p < 64;
            tab[(SITX + u) * 256] = ok ? (int)((py * ysy + px * ysx + view_chan_off(wp.dy, n0 + 4 * q)) * 4) : OOB;
            hyy[u >> 2] |= (unsigned)(ok ? (py | (px << 3)) : 0) << (8 * (u & 3));
        }
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
        const int ylo = max(0, 1 - it.y0), yhi = min(HH, wp.H + 1 - it.y0);
        const int xlo = max(0, 1 - it.x0), xhi = min(HW, wp.W + 1 - it.x0);
        int so[SITX], yo[SITY];
#pragma unroll
        for (int u = 0; u < SITX; ++u) so[u] = tab[u * 256];
#pragma unroll
        for (int u = 0; u < SITY; ++u) yo[u] = tab[(SITX + u) * 256];
        if (ylo | xlo | (yhi - HH) | (xhi - HW)) {                  // border tile groups (the empty asm keeps it a branch)
            asm volatile("" ::: "memory");
            const int ymax = wp.H - it.y0, xmax = wp.W - it.x0;
#pragma unroll
            for (int u = 0; u < SITX; ++u) {
                unsigned w = hyx[u >> 2];
                asm volatile("" : "+v"(w));
                const int b = (int)((w >> (8 * (u & 3))) & 255u);
                const int hy = b & 7, hx = b >> 3;
                so[u] = (hy >= ylo && hy < yhi && hx >= xlo && hx < xhi) ? so[u] : OOB;
            }
#pragma unroll
            for (int u = 0; u < SITY; ++u) {
                unsigned w = hyy[u >> 2];
                asm volatile("" : "+v"(w));
                const int b = (int)((w >> (8 * (u & 3))) & 255u);
                yo[u] = ((b & 7) < ymax && (b >> 3) < xmax) ? yo[u] : OOB;
            }
        }
        const long org = (long)((size_t)it.n * wp.x.nstride) + (long)(it.y0 - 1) * (long)isy + (long)(it.x0 - 1) * (long)isx;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(wp.x.p)) + org * 4, 0, 0x7fffff00, RSRC3);
        const size_t yorg = (size_t)it.n * wp.dy.nstride + it.y0 * ysy + it.x0 * ysx;
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(wp.dy.p)) + yorg * 4, 0, 0x7fffff00, RSRC3);
#if defined(__HIP_DEVICE_COMPILE__)                                   // (the host pass has no LDS address space to cast to)
        float* const dx = rawx + buf * (GM::RAWX * 4) + wave * 256;
        float* const dyp = rawy + buf * (GM::RAWY * 4) + wave * 256;
#pragma unroll
        for (int u = 0; u < SITX; ++u)
            if (4 * u + wave < NCHX) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dx + u * 1024), 16, so[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < SITY; ++u)
            if (4 * u + wave < NCHY) __builtin_amdgcn_raw_ptr_buffer_load_lds(ry, (lds_ptr_t)(dyp + u * 1024), 16, yo[u], 0, 0, 0);
#else
        (void)rs; (void)ry; (void)so; (void)yo; (void)buf;
#endif
    };

    // ---- the wave's rows of the input patch and of the 2 x 2 gradient block
    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int rb = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sg1 = wave == 1 ? 1.f : -1.f;
    const f32x2_t sg = {sg1, sg1};
    const int y_first = wave == 3 ? 1 : 0;                          // xi = 3: r = y1.
    const float cy1 = wave == 1 ? 1.f : (wave == 2 ? -1.f : 0.f);   // r = y(first). + cy y1.
    const f32x2_t cy = {cy1, cy1};
    const wino::PkConsts pkc = wino::pk_consts();
    // k-slots 0..3 take tiles tx0 + {0, 4, 1, 5}: the two k-slots of a 32-lane half read pixels 8 apart = 32 banks (mod 64) apart
    // at either pitch, which is what an 8-byte read per lane needs.  Channel blocks come in PAIRS: lane m of blocks (2 p, 2 p + 1)
    // holds channels 32 p + 2 m and 32 p + 2 m + 1 -- adjacent floats, ONE ds_read_b64 = the register pair the packed
    // instructions work on (the closing transform knows the permutation: wino_wgrad_finish_kernel, `paired`); an odd last block
    // holds channels 16 (K - 1) + m.
    const int txo = (lq & 1) * 4 + (lq >> 1);
    const int x_lane = (2 * txo * SPX) * 4, y_lane = (2 * txo * SPY) * 4;      // floats, relative to the step's first tile

    f32x4 acc[4][KQ][NT];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
        for (int i = 0; i < KQ; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[nu][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float dbacc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) dbacc[j] = 0.f;

    if (tg < tg_hi) {
        Item cur = decode(tg);
        stage_issue(cur, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int buf = 0;
        for (;;) {
            const int ntg = tg + nsub;
            const bool has_next = ntg < tg_hi;
            Item nxt = cur;
            const float* const bx = rawx + buf * (GM::RAWX * 4) + x_lane;
            const float* const by = rawy + buf * (GM::RAWY * 4) + y_lane;
            // raw values of k-step s.  Channel blocks in pairs (i, i + 1) = the two halves of a register pair; an odd last block alone.
            constexpr int KP = KQ / 2, NP = NT / 2;
            f32x2_t xa[KP ? KP : 1][4], xb[KP ? KP : 1][4], ya[NP ? NP : 1][2], yb[NP ? NP : 1][2];
            float xa1[4], xb1[4], ya1[2], yb1[2];                    // (KQ, NT odd)
            auto fetch = [&](int s) __attribute__((always_inline)) {
                const int ty = s >> 1, tx0 = 2 * (s & 1);
                const float* pa = bx + ((2 * ty + ra) * HW + 2 * tx0) * (SPX * 4);
                const float* pb = bx + ((2 * ty + rb) * HW + 2 * tx0) * (SPX * 4);
                const float* py = by + ((2 * ty + y_first) * 16 + 2 * tx0) * (SPY * 4);
                const float* p1 = by + ((2 * ty + 1) * 16 + 2 * tx0) * (SPY * 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
#pragma unroll
                    for (int i = 0; i < KP; ++i) {
                        xa[i][c] = *reinterpret_cast<const f32x2_t*>(pa + 32 * i + 2 * l15 + c * (SPX * 4));
                        xb[i][c] = *reinterpret_cast<const f32x2_t*>(pb + 32 * i + 2 * l15 + c * (SPX * 4));
                    }
                    if (KQ & 1) { xa1[c] = pa[16 * (KQ - 1) + l15 + c * (SPX * 4)]; xb1[c] = pb[16 * (KQ - 1) + l15 + c * (SPX * 4)]; }
                }
#pragma unroll
                for (int c = 0; c < 2; ++c) {
#pragma unroll
                    for (int j = 0; j < NP; ++j) {
                        ya[j][c] = *reinterpret_cast<const f32x2_t*>(py + 32 * j + 2 * l15 + c * (SPY * 4));
                        yb[j][c] = *reinterpret_cast<const f32x2_t*>(p1 + 32 * j + 2 * l15 + c * (SPY * 4));
                    }
                    if (NT & 1) { ya1[c] = py[16 * (NT - 1) + l15 + c * (SPY * 4)]; yb1[c] = p1[16 * (NT - 1) + l15 + c * (SPY * 4)]; }
                }
            };
            fetch(0);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                // operands of this k-step
                float A[4][KQ], B[4][NT];
#pragma unroll
                for (int i = 0; i < KP; ++i) {
                    f32x2_t t[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) t[c] = wino::pk_fma(xb[i][c], sg, xa[i][c]);
                    const f32x2_t v0 = wino::pk_sub(pkc, t[0], t[2]), v1 = wino::pk_add(t[1], t[2]);
                    const f32x2_t v2 = wino::pk_sub(pkc, t[2], t[1]), v3 = wino::pk_sub(pkc, t[1], t[3]);
                    A[0][2 * i] = v0[0]; A[0][2 * i + 1] = v0[1]; A[1][2 * i] = v1[0]; A[1][2 * i + 1] = v1[1];
                    A[2][2 * i] = v2[0]; A[2][2 * i + 1] = v2[1]; A[3][2 * i] = v3[0]; A[3][2 * i + 1] = v3[1];
                }
                if (KQ & 1) {
                    float t[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) t[c] = fmaf(xb1[c], sg1, xa1[c]);
                    A[0][KQ - 1] = t[0] - t[2]; A[1][KQ - 1] = t[1] + t[2]; A[2][KQ - 1] = t[2] - t[1]; A[3][KQ - 1] = t[1] - t[3];
                }
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const f32x2_t r0 = wino::pk_fma(yb[j][0], cy, ya[j][0]), r1 = wino::pk_fma(yb[j][1], cy, ya[j][1]);
                    const f32x2_t m1 = wino::pk_add(r0, r1), m2 = wino::pk_sub(pkc, r0, r1);
                    B[0][2 * j] = r0[0]; B[0][2 * j + 1] = r0[1]; B[1][2 * j] = m1[0]; B[1][2 * j + 1] = m1[1];
                    B[2][2 * j] = m2[0]; B[2][2 * j + 1] = m2[1]; B[3][2 * j] = r1[0]; B[3][2 * j + 1] = r1[1];
                    dbacc[2 * j] += m1[0]; dbacc[2 * j + 1] += m1[1];          // (wave 1: the tile's four pixels; the other waves' sums are unused)
                }
                if (NT & 1) {
                    const float r0 = fmaf(yb1[0], cy1, ya1[0]), r1 = fmaf(yb1[1], cy1, ya1[1]);
                    B[0][NT - 1] = r0; B[1][NT - 1] = r0 + r1; B[2][NT - 1] = r0 - r1; B[3][NT - 1] = r1;
                    dbacc[NT - 1] += r0 + r1;
                }
                __builtin_amdgcn_sched_barrier(0);
                auto mfmas = [&](int nu) __attribute__((always_inline)) {
#pragma unroll
                    for (int i = 0; i < KQ; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[nu][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[nu][i], B[nu][j], acc[nu][i][j], 0, 0, 0);
                };
                mfmas(0);
                mfmas(1);
                __builtin_amdgcn_sched_barrier(0);
                // the next k-step's raw values are requested once half of this step's operands are dead (register budget of <3,3>)
                if (s < 3) fetch(s + 1);
                if (s == 0 && has_next) {
                    // the next raw blocks: every wave left the buffers they go to before the barrier that ended the last iteration
                    nxt = decode(ntg);
                    stage_issue(nxt, buf ^ 1);
                }
                mfmas(2);
                mfmas(3);
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's pieces of the next blocks have landed
            __syncthreads();
            if (!has_next) break;
            cur = nxt;
            tg = ntg;
            buf ^= 1;
        }
    }
    // ---- the workgroup's share of dU: lane (column l15 = cout, rows 4 lq + r = cin) -> [wave][b][cin block][cout block][lane][4].
    // The COLUMN half of G^T . G is applied here, in the wave's registers (round 6): 12 planes leave instead of 16 -- a quarter of the slab bytes
    // (512 slabs of 147 KB per 48 x 48 weight gradient, written here and read by the first slab sum).  u[b] = sum over nu of G[nu][b] s(nu) dU[xi][nu]
    // with s(3) = -1 (the rows of A stored positive, see the header); the row half and row 3's sign are the closing kernel's.
    // (Measured and dropped: the row half here too -- the waves exchanging u[b] through LDS, one b at a time, 9 planes of dW[a][b] leaving:
    //  wino_wgrad_finish 0.175 -> 0.162 ms per cfg2 step, this kernel's launches + 0.010-0.015 ms for their three barrier pairs: nothing.)
#pragma unroll
    for (int i = 0; i < KQ; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const f32x4 h = .5f * (acc[1][i][j] + acc[2][i][j]);
            const f32x4 u[3] = {acc[0][i][j] + h, .5f * (acc[1][i][j] - acc[2][i][j]), h - acc[3][i][j]};
#pragma unroll
            for (int b = 0; b < 3; ++b)
                *reinterpret_cast<f32x4*>(myslab + ((size_t)((wave * 3 + b) * KQ + i) * NT + j) * 256 + lane * 4) = u[b];
        }
    // ---- bias gradient: wave 1's lanes hold, per cout 16 j + l15, the sum over the tiles of their k-slot
    __syncthreads();
    if (wave == 1) {
#pragma unroll
        for (int j = 0; j < NT; ++j) lds[(j * 4 + lq) * 16 + l15] = dbacc[j];
    }
    __syncthreads();
    if (tid < CO) {
        const int j = tid >> 4, c = tid & 15;
        const int co = j < 2 * (NT / 2) ? 32 * (j >> 1) + 2 * c + (j & 1) : 16 * j + c;        // (the lane's output channel: paired blocks)
        myslab[GM::FR + co] = (lds[(j * 4 + 0) * 16 + c] + lds[(j * 4 + 1) * 16 + c]) + (lds[(j * 4 + 2) * 16 + c] + lds[(j * 4 + 3) * 16 + c]);
    }
}

// out[g][i] = sum over the slabs k of group g (blockIdx.y: slabs [g per, (g + 1) per)) of slab[k * n + i], in the order of k
// (bitwise reproducible); n % 4 == 0.  The closing transform adds the groups, again in a fixed order: with two workgroups per CU
// there are 512 slabs, and one thread walking all of them serially was 25 us of latency per weight gradient.
__global__ void __launch_bounds__(256) wino_slab_sum_kernel(const float* __restrict__ slab, float* __restrict__ out, int n4, int nslabs_all,
                                                            int per) {
    const int k_lo = blockIdx.y * per;
    const int nslabs = min(per, nslabs_all - k_lo);
    const f32x4* s4 = reinterpret_cast<const f32x4*>(slab) + (size_t)k_lo * n4;
    f32x4* o4 = reinterpret_cast<f32x4*>(out) + (size_t)blockIdx.y * n4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
        f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
        int k = 0;
        for (; k + 8 <= nslabs; k += 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = s4[(size_t)(k + u) * n4 + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += v[u];
        }
        for (; k < nslabs; ++k) a += s4[(size_t)k * n4 + i];
        o4[i] = a;
    }
}

// dW[a][b][cin][cout] (+)= sum over xi of G[xi][a] sigma(xi) S[xi][b][cin][cout], sigma(3) = -1 (S: the slabs' sum, columns transformed);
// thread = (pair, cin block, cout block, lane): four input channels (rows 4 lq + r) of one output channel
__global__ void __launch_bounds__(256) wino_wgrad_finish_kernel(const float* __restrict__ S, float* __restrict__ dw, float* __restrict__ db,
                                                                int Cin, int Cout, int KQ, int NT, int ncin, int ncout, int accumulate,
                                                                int accumulate_db, int ngroups, size_t gstride, int paired) {
    const int F = 3 * KQ * NT, ST = 4 * F * 256 + 16 * NT;
    const int per_pair = KQ * NT * 64;
    const int total = ncin * ncout * per_pair;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int lane = idx & 63;
        int r = idx >> 6;
        const int j = r % NT; r /= NT;
        const int i = r % KQ; r /= KQ;
        const int pc = r;
        // second form of the kernel (`paired`): row m of the blocks (2 p, 2 p + 1) is channel 32 p + 2 m (+ 1), see its fetch
        const bool pi = paired && i < 2 * (KQ / 2), pj = paired && j < 2 * (NT / 2);
        const int cin_b = (pc / ncout) * 16 * KQ + (pi ? 32 * (i >> 1) + 8 * (lane >> 4) + (i & 1) : 16 * i + 4 * (lane >> 4));
        const int cin_st = pi ? 2 : 1;
        const int co = (pc % ncout) * 16 * NT + (pj ? 32 * (j >> 1) + 2 * (lane & 15) + (j & 1) : 16 * j + (lane & 15));
        if (co >= Cout) continue;
        // the slabs hold u[xi][b] (columns already transformed by the weight-gradient kernel); rows: sigma(3) = -1, then G^T
        f32x4 u[4][3];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const float* q = S + (size_t)pc * ST + ((size_t)((xi * 3 + b) * KQ + i) * NT + j) * 256 + lane * 4;
                f32x4 v = *reinterpret_cast<const f32x4*>(q);
                for (int g = 1; g < ngroups; ++g) v += *reinterpret_cast<const f32x4*>(q + g * gstride);
                u[xi][b] = xi == 3 ? -v : v;
            }
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const f32x4 h = .5f * (u[1][b] + u[2][b]);
            const f32x4 g[3] = {u[0][b] + h, .5f * (u[1][b] - u[2][b]), u[3][b] + h};
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int cin = cin_b + rr * cin_st;
                    if (cin < Cin) {
                        float* p = dw + ((size_t)((a * 3 + b) * Cin + cin)) * Cout + co;
                        *p = accumulate ? *p + g[a][rr] : g[a][rr];
                    }
                }
        }
    }
    if (db) {
        for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < Cout; c += gridDim.x * blockDim.x) {
            const int pcq = c / (16 * NT);                          // (cin chunk 0)
            const float* q = S + (size_t)pcq * ST + 4 * F * 256 + (c - pcq * 16 * NT);
            float v = *q;
            for (int g = 1; g < ngroups; ++g) v += q[g * gstride];
            db[c] = accumulate_db ? db[c] + v : v;
        }
    }
}

int wgrad_cu_count() {
    static const int n = [] {
        int dev = 0, v = 0;
        HIP_CHECK(hipGetDevice(&dev));
        HIP_CHECK(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
        return v;
    }();
    return n;
}

struct WgScratch { hipStream_t stream; float* buf; size_t floats; };
float* wgrad_scratch(hipStream_t s, size_t floats) {       // grow-only, one buffer per stream (launches on a stream are ordered)
    static std::mutex mu;
    static std::vector<WgScratch> all;
    std::lock_guard<std::mutex> lk(mu);
    for (auto& e : all) {
        if (e.stream != s) continue;
        if (e.floats < floats) {
            HIP_CHECK(hipStreamSynchronize(s));
            HIP_CHECK(hipFree(e.buf));
            HIP_CHECK(hipMalloc((void**)&e.buf, floats * sizeof(float)));
            e.floats = floats;
        }
        return e.buf;
    }
    WgScratch e{s, nullptr, floats};
    HIP_CHECK(hipMalloc((void**)&e.buf, e.floats * sizeof(float)));
    all.push_back(e);
    return e.buf;
}

template <int KQ, int NT>
void launch_wgrad2(hipStream_t s, WinoWgradParams& wp, int SX) {
    typedef Wg2Geom<KQ, NT> GM;
    static std::once_flag once;
    std::call_once(once, [&]() {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_wgrad2_kernel<KQ, NT>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)GM::LDS_BYTES));
    });
    DL4DS_LAUNCH((conv_wino_wgrad2_kernel<KQ, NT>), dim3(8 * SX), dim3(256), GM::LDS_BYTES, s, wp);
    HIP_CHECK(hipGetLastError());
}

}  // namespace

// 3x3, stride 1, SAME: dw = [3][3][Cin][Cout], db = [Cout] or null.  false = not eligible (the caller runs the direct kernels).
bool conv2d_wino_wgrad(hipStream_t s, const TView& x, const TView& dy, float* dw, int accumulate, float* db, int accumulate_db) {
    if (getenv("DL4DS_NO_WINOGRAD") || getenv("DL4DS_NO_WINOGRAD_WGRAD")) return false;
    const char* force = test_env("DL4DS_WINO_FORCE");
    if (x.sc || dy.sc || !x.vec || !dy.vec || (x.C & 3) || (dy.C & 3)) return false;
    if (x.C < 24 || dy.C < 24) return false;
    // (measured, B = 64 at 128^2 / 256^2: 48 -> 48 0.31 vs 0.39 ms direct, 48 -> 192 1.10 vs 1.28, 40 -> 48 0.30 vs 0.39; chunks of 32
    //  channels on either side do not pay: 48 -> 32 0.90 vs 0.88, 32 -> 32 0.19 vs 0.17)
    // (first form, B = 64: chunks of 32 channels on either side did not pay -- 48 -> 32 at 256^2 0.90 vs 0.88 ms direct, 32 -> 32 0.19
    //  vs 0.17; second form: 0.58 vs 0.84 and 0.127 vs 0.152, so every layer with >= 24 channels on both sides takes it)
    auto span = [](const TView& v) { const size_t r = std::max(v.d2s, 1); return (size_t)8 * v.W * r * r * v.ld * 4; };
    if (span(x) >= (1ull << 31) || span(dy) >= (1ull << 31)) return false;
    const int KQ = (cdiv(x.C, 32) * 32 < cdiv(x.C, 48) * 48) ? 2 : 3;
    const int NT = (cdiv(dy.C, 32) * 32 < cdiv(dy.C, 48) * 48) ? 2 : 3;
    WinoWgradParams wp;
    wp.x = x; wp.dy = dy;
    wp.Cin = x.C; wp.Cout = dy.C; wp.H = x.H; wp.W = x.W;
    wp.ncin = cdiv(x.C, 16 * KQ);
    wp.ncout = cdiv(dy.C, 16 * NT);
    wp.tgx = cdiv(x.W, 16);
    wp.tgy = cdiv(x.H, 4);
    wp.m_tgx = div_magic(wp.tgx);
    wp.m_tgy = div_magic(wp.tgy);
    const long ntg = (long)wp.tgx * wp.tgy * x.N;
    if (ntg >= (1l << 20)) return false;
    wp.ntg = (int)ntg;
    wp.per_xcd = cdiv(wp.ntg, 8);
    const int npair = wp.ncin * wp.ncout;
    const int SXmax = std::max(2 * wgrad_cu_count() / 8, 1);          // two workgroups per CU
    if (npair > SXmax) return false;
    int SX = (SXmax / npair) * npair;
    if (force && atoi(force) > 0) SX = std::min(SX, atoi(force) * npair);
    const int nsub = SX / npair;
    // fewer than 8 tile groups per workgroup: the slabs dominate.  (Round 6, cfg2 at per-GPU batch 16 = 8 groups per workgroup: the
    // former limit of 16 sent the five 48-channel layers to the direct kernels -- weight-gradient family 1.46 -> 1.35 ms per step of 16,
    // 3 845 -> 3 950 samples/s; one workgroup per CU with 16 groups each is slower (3 640), and so is a limit of 4 for the 32-channel shapes)
    if (!force && (long)wp.per_xcd < 8l * nsub) return false;
    const int F = 3 * KQ * NT, ST = 4 * F * 256 + 16 * NT;
    const int nslabs = 8 * nsub;
    const size_t per_k = (size_t)npair * ST;
    const int ngroups = std::min(nslabs, 16), per_group = cdiv(nslabs, ngroups);
    float* const slab = wgrad_scratch(s, per_k * (nslabs + ngroups + 1));
    float* const part = slab + per_k * nslabs;             // one partial sum per group of slabs,
    float* const sum = part + per_k * ngroups;             // then their sum (two short launches instead of one long serial walk)
    wp.slab = slab;
    const double px = (double)x.N * x.H * x.W;
    {
        const double issued = 2.0 * (px / 4.0) * 16.0 * (16.0 * KQ * wp.ncin) * (16.0 * NT * wp.ncout) +
                              (px / 4.0) * (32.0 * x.C * wp.ncout + 12.0 * dy.C * wp.ncin);
        ProfScope ps(s, "conv_wino_wgrad<" + std::to_string(KQ) + "," + std::to_string(NT) + ">", issued,
                     4.0 * (px * (x.C + dy.C) + 9.0 * x.C * dy.C), 2.0 * px * 9.0 * x.C * dy.C);
        if (KQ == 2) {
            if (NT == 2) launch_wgrad2<2, 2>(s, wp, SX); else launch_wgrad2<2, 3>(s, wp, SX);
        } else {
            if (NT == 2) launch_wgrad2<3, 2>(s, wp, SX); else launch_wgrad2<3, 3>(s, wp, SX);
        }
    }
    ProfScope ps(s, "wino_wgrad_finish", 0.0, 4.0 * (double)per_k * (nslabs + 2));
    const int n4 = (int)(per_k / 4);
    const int ng = cdiv(nslabs, per_group);
    DL4DS_LAUNCH(wino_slab_sum_kernel, dim3(std::min(cdiv(n4, 256), 2048), ng), dim3(256), 0, s, slab, part, n4, nslabs, per_group);
    HIP_CHECK(hipGetLastError());
    const int total = npair * KQ * NT * 64;
    // (round 5, measured and dropped: the closing transform adding the <= 16 group sums itself -- ngroups = ng on `part` -- instead of
    //  this second short launch: wino_wgrad_finish 0.20 -> 0.78 ms per cfg2 step, 5 020 -> 4 790 samples/s; its 16 x 16 strided
    //  float4 loads per thread are far slower than the coalesced sum)
    // (round 6, measured and dropped: the same with a closing kernel built for it -- one 1024-thread workgroup per (pair, cin block, cout block),
    //  thread (position, lane) adding the 16 group sums of its float4 coalesced, LDS handing a lane's 16 positions to the transform threads:
    //  same bits, one launch less, and 0.1 % SLOWER at B = 64, 0.3-0.7 % at B = 16 in three alternating runs -- 9 workgroups take as long as the two
    //  short launches; the 20 us per weight gradient are the first sum's 75 MB)
    DL4DS_LAUNCH(wino_slab_sum_kernel, dim3(std::min(cdiv(n4, 256), 2048), 1), dim3(256), 0, s, part, sum, n4, ng, ng);
    HIP_CHECK(hipGetLastError());
    DL4DS_LAUNCH(wino_wgrad_finish_kernel, dim3(std::min(cdiv(total, 256), 1024)), dim3(256), 0, s, sum, dw, db, x.C, dy.C, KQ, NT,
                       wp.ncin, wp.ncout, accumulate, accumulate_db, 1, per_k, 1);
    HIP_CHECK(hipGetLastError());
    return true;
}
