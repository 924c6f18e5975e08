// dl4ds_amd -- Winograd F(4x4, 3x3) form of the MFMA-bound 3x3 convolutions (round 6; forward and, on the transposed filter, dgrad).
//
// Y = A^T [ (G g G^T) . (B^T d B) ] A on 6 x 6 input patches / 4 x 4 output tiles (points 0, +-1, +-2, inf): 36 multiplications per
// 16 outputs and (cin, cout) pair = 2.25 per output, against 4 of F(2x2, 3x3) (conv_wino_kernel.h) and 9 of the direct form
// (blocks.py:210-230, 433-454).  What decides the structure on gfx950 is where 36 transformed filter positions live: the fp32 MFMA
// runs at the vector rate (rule 1), so a filter fragment re-read from LDS per MFMA costs what the
// transform saves -- the filter has to be register-stationary as in the F(2x2) kernels.  36 positions x (16 KQ x 16 NT) / 64 lanes
// is 36 KQ NT registers per workgroup lane: ONE workgroup of four waves per CU (512 registers per lane each), wave (a, b) owning the
// 3 x 3 block of positions xi in 3a..3a+2, nu in 3b..3b+2 -- 9 KQ NT 4 filter registers + 36 NT accumulators, which fits for
// (KQ, NT) = (3, 2), (2, 2), (2, 3) and does not for (3, 3).
//   * tile group = 4 x 4 tiles = 16 x 16 output pixels (MFMA column l15 = tile (l15 >> 2, l15 & 3)), halo 18 x 18;
//   * the halo is staged per 16-channel chunk (18 x 18 x 64 B, DMA'd with buffer_load ... lds) into a ring of chunk buffers; chunk
//     kq of the NEXT tile group is requested as soon as its buffer's last reader has passed a barrier (one barrier per chunk), so the
//     loads of a tile group are spread over the whole previous one (one workgroup per CU: nobody else hides them);
//   * per chunk, lane (tile, channel quad lq) reads 5 x 5 float4 of its patch and computes ITS block of B^T d B in registers:
//     rows first (three xi of five columns), then columns (three nu of three xi) -- both by the same wave-uniform recipe
//       p = R0 + c1 R1, q = R2 + c1 R3, (q + kp p, q - kp p), s4 R4 + s1 R1 + s0 R0
//     on rows / columns taken in a block-dependent ORDER (block 0: 4,2,3,1,0; block 1: 3,1,4,2,5), so that one instruction
//     stream serves all four waves; the sign of the one output that comes out negated (index 2) is folded into the filter;
//   * 9 positions x 4 k-steps x NT MFMAs (16x16x4 f32) per chunk into 9 NT accumulators;
//   * after the K loop the 36 NT products of a tile go to LDS, and every thread finishes (tile, cout quad, two of the tile's four
//     columns): A^T M A from 30 float4, epilogue (bias, residual, ReLU, mask, accumulation: the forms of the F(2x2) kernel), 8 stores.
// LDS pitches: a chunk pixel = 5 sixteen-byte slots (4 quads + 1), halo rows 92 slots with rows of odd tile rows displaced by 2
// slots, so that the 16 lanes the hardware serves together ({tile rows 0, 3 at quad q} + {tile rows 1, 2 at quad q + 1}) cover all
// 16 slot residues; products at a pitch of 16 NT + 8 floats (2 mod 4 slots).
// Numerics: fp32 throughout; forward error 4.5e-6 of the layer's largest output on 48 -> 48 (F(2x2): 2.7e-7, direct 1.6e-7;
// tools/f44_accuracy.py), from the transform constants 4, 5, 8 and 1/24.
#pragma once
#include "conv_wino_kernel.h"

namespace wino4 {
using wino::f32x2;
using wino::i32x4_t;
using wino::view_strides;

template <int KQ, int NT>
struct Geom {
    static constexpr int Q4 = 4 * KQ, CO = 16 * NT, NQ = 4 * NT;
    static constexpr int HP = 18;                                    // halo: 18 x 18 pixels
    static constexpr int SP = 5;                                     // chunk pixel pitch in 16-byte slots (4 quads + 1)
    static constexpr int ROWS = HP * SP + 2;                         // slots per halo row (+2: the displacement of odd tile rows)
    static constexpr int NCHP = (HP * ROWS + 63) / 64;               // DMA pieces (64 slots) per chunk
    static constexpr int CHS = NCHP * 64;                            // slots per chunk buffer
    static constexpr int SIT = (NCHP + 3) / 4;                       // pieces per wave
    static constexpr int PP = CO + 8;                                // pitch of the products (floats)
    static constexpr int P = 36 * 16 * PP;                           // floats
    // chunk buffers 0, 1 are their own; chunk 2 (KQ = 3) is staged where the products go (free while the K loop runs)
    static constexpr int NBUF = KQ < 3 ? KQ : 2;
    static constexpr int CTAB = 16;                                  // ints: byte offset of channel quad (kq, q) in the input view
    static constexpr int TAB = (SIT + (SIT + 2) / 3) * 256;          // per-thread DMA offsets and packed halo coordinates (ints)
    static constexpr size_t LDS_BYTES = (size_t)(NBUF * CHS * 4 + P + CTAB + TAB) * 4;
    static_assert(CHS * 4 <= P, "chunk 2 overlays the products");
    static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
    static constexpr int F = 36 * KQ * NT;                           // filter registers per lane
    static_assert(F + 36 * NT <= 432, "filter + accumulators leave room for the transforms");
};

// the block's order of rows / columns and its recipe constants (see the header comment)
__device__ __forceinline__ int role_index(int blk, int k) {
    return blk == 0 ? (k == 0 ? 4 : k == 1 ? 2 : k == 2 ? 3 : k == 3 ? 1 : 0) : (k == 0 ? 3 : k == 1 ? 1 : k == 2 ? 4 : k == 3 ? 2 : 5);
}
struct Recipe { f32x2 c1, kp, s4, s1, s0; };
__device__ __forceinline__ Recipe recipe_of(int blk) {
    float c1 = blk ? -1.f : -4.f, kp = blk ? 2.f : 1.f, s4 = blk ? 1.f : 4.f, s1 = blk ? 4.f : -5.f, s0 = blk ? -5.f : 1.f;
    // (opaque and in scalar registers -- blk is wave-uniform: the constants stay operands of packed multiply-adds)
    asm volatile("" : "+v"(c1), "+v"(kp), "+v"(s4), "+v"(s1), "+v"(s0));
    Recipe r;
    r.c1 = (f32x2){c1, c1}; r.kp = (f32x2){kp, kp};
    r.s4 = (f32x2){s4, s4}; r.s1 = (f32x2){s1, s1}; r.s0 = (f32x2){s0, s0};
    return r;
}
__device__ __forceinline__ f32x4 fma4v(const f32x4 b, const f32x2 s, const f32x4 c) {          // c + s * b
    return wino::cat4(wino::pk_fma(wino::lo2(b), s, wino::lo2(c)), wino::pk_fma(wino::hi2(b), s, wino::hi2(c)));
}
__device__ __forceinline__ f32x4 mul4v(const f32x4 b, const f32x2 s) {
    return wino::cat4(wino::lo2(b) * s, wino::hi2(b) * s);
}
// five values in the block's order -> (single, q + kp p, q - kp p)
__device__ __forceinline__ void transform3(const Recipe& rc, const f32x4 R0, const f32x4 R1, const f32x4 R2, const f32x4 R3, const f32x4 R4,
                                           f32x4& o0, f32x4& o1, f32x4& o2) {
    const f32x4 p = fma4v(R1, rc.c1, R0);
    const f32x4 q = fma4v(R3, rc.c1, R2);
    o1 = fma4v(p, rc.kp, q);
    o2 = fma4v(-p, rc.kp, q);                                       // (the negation is a source modifier)
    o0 = fma4v(R0, rc.s0, fma4v(R1, rc.s1, mul4v(R4, rc.s4)));
}

// workgroup barrier for LDS traffic only: __syncthreads() also waits for every outstanding vector-memory operation (vmcnt(0)) -- here
// the DMA pieces just requested for the NEXT tile group, whose whole latency it would expose
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int KQ, int NT, int EPI>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) conv_wino4_kernel(const WinoParams wp) {
    typedef Geom<KQ, NT> GM;
    const ConvParams& a = wp.c;
    constexpr int CO = GM::CO, NQ = GM::NQ, SP = GM::SP, ROWS = GM::ROWS, PP = GM::PP, HP = GM::HP, SIT = GM::SIT, NCHP = GM::NCHP;
    constexpr int OOB = (int)0xffffff00u;
    constexpr int RSRC3 = 0x00020000;
    constexpr bool OLDF = (EPI & WINO_OLDF) != 0, ADD = (EPI & WINO_ADD) != 0, MASK = (EPI & WINO_MASK) != 0, OLDA = (EPI & WINO_OLDA) != 0;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const Pb = lds + GM::NBUF * GM::CHS * 4;
    int* const ctab = reinterpret_cast<int*>(Pb + GM::P);
    auto chunk_buf = [&](int kq) { return kq < GM::NBUF ? lds + kq * (GM::CHS * 4) : Pb; };

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, lq = lane >> 4;
    const int blk_a = wave >> 1, blk_b = wave & 1;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, SX = gridDim.x >> 3;
    const int nsub = SX / wp.nchunk;
    const int chunk = slot % wp.nchunk, sub = slot / wp.nchunk;
    const int n0 = chunk * CO;
    const int tg_lo = xcd * wp.per_xcd, tg_hi = min(wp.ntg, tg_lo + wp.per_xcd);
    int tg = tg_lo + sub;
    if (sub >= nsub || tg >= tg_hi) return;                         // (whole workgroup)

    // ---- staging: DMA piece ch = 4 u + wave fills slots [64 ch, 64 ch + 64) of a chunk buffer; slot L = row y, then (x, quad q)
    size_t isy, isx;
    view_strides(a.in, isy, isx);
    if (tid < 4 * KQ) {
        const int cq = wp.cin0 + 4 * tid;
        ctab[tid] = cq < a.Cin ? (int)(view_chan_off(a.in, cq) * 4) : OOB;
    }
    // tab[u][tid]: byte offset of the slot's pixel | its quad (bits 0-1), OOB for a pad slot; then (y | x << 5) of every piece's
    // slot, ten bits each (only border groups look) -- in LDS: ten registers the K loop has no room for
    int* const tab = ctab + GM::CTAB + tid;
    {
        unsigned hyx[(SIT + 2) / 3];
#pragma unroll
        for (int u = 0; u < (SIT + 2) / 3; ++u) hyx[u] = 0;
#pragma unroll
        for (int u = 0; u < SIT; ++u) {
            const int L = (4 * u + wave) * 64 + lane;
            const int y = L / ROWS;
            const int rem = L - y * ROWS - 2 * ((y >> 2) & 1);
            const int x = rem / SP, q = rem - x * SP;
            const bool ok = y < HP && rem >= 0 && rem < HP * SP && q < 4;
            tab[u * 256] = ok ? ((int)((y * isy + x * isx) * 4) | q) : OOB;
            hyx[u / 3] |= (unsigned)(ok ? (y | (x << 5)) : 0) << (10 * (u % 3));
        }
#pragma unroll
        for (int u = 0; u < (SIT + 2) / 3; ++u) tab[(SIT + u) * 256] = (int)hyx[u];
    }
    struct Item { int n, y0, x0; };
    auto decode = [&](int t) {
        const int q = fast_div(t, wp.m_tgx);
        const int bx = t - q * wp.tgx;
        const int n = fast_div(q, wp.m_tgy);
        const int by = q - n * wp.tgy;
        Item it;
        it.n = n; it.y0 = by * 16; it.x0 = bx * 16;
        return it;
    };
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    auto stage_issue = [&](const Item& it, int kq) __attribute__((always_inline)) {
        const int ylo = max(0, 1 - it.y0), yhi = min(HP, a.H + 1 - it.y0);
        const int xlo = max(0, 1 - it.x0), xhi = min(HP, a.W + 1 - it.x0);
        const bool border = (ylo | xlo | (yhi - HP) | (xhi - HP)) != 0;
        int so[SIT];
#pragma unroll
        for (int u = 0; u < SIT; ++u) {
            const int pv = tab[u * 256];
            const int co = ctab[4 * kq + (pv & 3)];
            bool ok = pv != OOB && co != OOB;
            if (border) {
                const int b = (int)(((unsigned)tab[(SIT + u / 3) * 256] >> (10 * (u % 3))) & 1023u);
                const int hy = b & 31, hx = b >> 5;
                ok = ok && hy >= ylo && hy < yhi && hx >= xlo && hx < xhi;
            }
            so[u] = ok ? (pv & ~3) + co : OOB;
        }
        const long org = (long)((size_t)it.n * a.in.nstride) + (long)(it.y0 - 1) * (long)isy + (long)(it.x0 - 1) * (long)isx;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(a.in.p)) + org * 4, 0, 0x7fffff00, RSRC3);
#if defined(__HIP_DEVICE_COMPILE__)                                   // (the host pass has no LDS address space to cast to)
        float* const dst = chunk_buf(kq) + wave * 256;
#pragma unroll
        for (int u = 0; u < SIT; ++u)
            if (4 * u + wave < NCHP)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + u * 1024), 16, so[u], 0, 0, 0);
#else
        (void)rs; (void)so;
#endif
    };

    // ---- K loop operands: the lane's 25 patch addresses (floats from a chunk buffer's start), rows and columns in block order
    const int k_ty = l15 >> 2, k_tx = l15 & 3;
    // (column roles 0, 1 are columns X, X - 2 and roles 2, 3 columns Y, Y - 2 in both blocks: three column bases, the rest immediates)
    int kaddr[5][3];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int row = 4 * k_ty + role_index(blk_a, k);
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) {
            const int col = 4 * k_tx + role_index(blk_b, c3 == 0 ? 1 : (c3 == 1 ? 3 : 4));
            kaddr[k][c3] = (row * ROWS + 2 * ((row >> 2) & 1) + col * SP + lq) * 4;
        }
    }
    auto patch = [&](const float* rb, int k, int c) __attribute__((always_inline)) {
        // c = 0, 2: two columns right of c = 1, 3
        const int c3 = c >> 1, plus = (c == 0 || c == 2) ? 2 * SP * 4 : 0;
        return *reinterpret_cast<const f32x4*>(rb + kaddr[k][c3] + plus);
    };
    const Recipe rca = recipe_of(blk_a), rcb = recipe_of(blk_b);
    const wino::PkConsts pkc = wino::pk_consts();

    // ---- phase C: thread finishes items e = tid + 256 u = (tile e & 15, cout quad (e >> 4) % NQ, column pair (e >> 4) / NQ)
    static_assert(32 * NQ == 256, "phase C: one item per thread (NT = 2)");
    size_t osy, osx;
    view_strides(a.out, osy, osx);
    const int nq = min(NQ, max(0, (a.Cout - n0) >> 2));
    const float floor_v = a.relu ? 0.f : -3.0e38f;
    const bool want_bias = wp.first && a.bias != nullptr;
    const int bias_max = max(a.Cout - 4, 0);

    // ---- the wave's block of the transformed filter, as MFMA first operands: lane (row l15, k-slot lq), k-step ks = 4 kq + s
    //      -> U[x][n][cin = cin0 + 16 kq + 4 lq + s][cout = n0 + 16 cb + l15] of position (xi(a, x), nu(b, n)), signs folded
    //      (wino4_filter_kernel left them in exactly this order)
    constexpr int F = GM::F;
    float U[3][3][4 * KQ][NT];
    {
        const f32x4* up = reinterpret_cast<const f32x4*>(wp.u) + ((size_t)(chunk * 4 + wave) * (F / 4)) * 64 + lane;
#pragma unroll
        for (int f4 = 0; f4 < F / 4; ++f4) {
            f32x4 v = up[f4 * 64];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f = 4 * f4 + j;
                float x = v[j];
                asm volatile("" : "+a"(x));                          // (accumulator-file registers: MFMA operands straight from there)
                U[f / (12 * KQ * NT)][(f / (4 * KQ * NT)) % 3][(f / NT) % (4 * KQ)][f % NT] = x;
            }
        }
    }

    __syncthreads();                                                // ctab
#ifdef WINO_TRACE
    unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tr_t = clock64();
#endif
    Item cur = decode(tg);
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq) stage_issue(cur, kq);
    const bool s_full = 4 * (SIT - 1) + wave < NCHP;               // this wave requests SIT pieces per chunk (else SIT - 1)
    bool first_it = true;
    WT(0);
    for (;;) {
        const int ntg = tg + nsub;
        const bool has_next = ntg < tg_hi;
        Item nxt = cur;
        if (has_next) nxt = decode(ntg);
        f32x4 acc[3][3][NT];
#pragma unroll
        for (int x = 0; x < 3; ++x)
#pragma unroll
            for (int n = 0; n < 3; ++n)
#pragma unroll
                for (int cb = 0; cb < NT; ++cb) acc[x][n][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq) {
            // this wave's pieces of chunk kq have landed: everything but the requests made after it may still be in flight (its
            // vector-memory operations complete in order) -- the last chunk's request is followed by the 8 stores of phase C, the
            // others by a later chunk's S pieces and those stores; the first tile group's three requests are simply all awaited
            if (first_it) wait_vm<0>();
            else if (kq == KQ - 1) wait_vm<8>();
            else if (s_full) wait_vm<GM::SIT + 8>();
            else wait_vm<GM::SIT - 1 + 8>();
            lds_barrier();                                        // ... everybody's; and every wave has left chunk kq - 1
            WT(1);
            // the buffer of chunk kq - 1 is free: the next tile group's chunk kq - 1 goes there (chunk 2 waits for the products)
            if (kq >= 1 && kq - 1 < GM::NBUF && has_next) stage_issue(nxt, kq - 1);
            WT(2);
            const float* const rb = chunk_buf(kq);
            f32x4 T[3][5];
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                f32x4 R[5];
#pragma unroll
                for (int k = 0; k < 5; ++k) R[k] = patch(rb, k, c);
                transform3(rca, R[0], R[1], R[2], R[3], R[4], T[0][c], T[1][c], T[2][c]);
            }
#pragma unroll
            for (int x = 0; x < 3; ++x) {
                f32x4 V[3];
                transform3(rcb, T[x][0], T[x][1], T[x][2], T[x][3], T[x][4], V[0], V[1], V[2]);
#pragma unroll
                for (int n = 0; n < 3; ++n)
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                        for (int cb = 0; cb < NT; ++cb)
                            acc[x][n][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[x][n][4 * kq + s4][cb], V[n][s4], acc[x][n][cb], 0, 0, 0);
            }
        }
        WT(3);
        lds_barrier();                                            // every wave has left the last chunk (its buffer may be the products')
        if (has_next && KQ - 1 < GM::NBUF) stage_issue(nxt, KQ - 1);
        // ---- the products of position (xi, nu), tile l15, couts 16 cb + 4 lq ..
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            const int xi = blk_a == 0 ? x : (x == 0 ? 5 : x + 2);
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                const int nu = blk_b == 0 ? n : (n == 0 ? 5 : n + 2);
                float* const pw = Pb + ((xi * 6 + nu) * 16 + l15) * PP + 4 * lq;
#pragma unroll
                for (int cb = 0; cb < NT; ++cb) *reinterpret_cast<f32x4*>(pw + 16 * cb) = acc[x][n][cb];
            }
        }
        lds_barrier();                                            // products complete
        WT(4);
        // ---- C: Y = A^T M A for (tile, quad, two columns), epilogue, store
        {
            const size_t pb_ = cur.y0 * osy + cur.x0 * osx;
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<char*>(a.out.p) + ((size_t)cur.n * a.out.nstride + pb_) * 4, 0, 0x7fffff00, RSRC3);
            const int tile = tid & 15, rest = tid >> 4;
            const int jh = rest / NQ, quad = rest - jh * NQ;        // (NQ = 8: jh is wave-uniform)
            const int ty = tile >> 2, tx = tile & 3;
            const int py0 = 4 * ty, px0 = 4 * tx + 2 * jh;
            const bool q_ok = quad < nq;
            const int obase = (int)((py0 * osy + px0 * osx + view_chan_off(a.out, min(n0 + 4 * quad, max(a.Cout - 4, 0)))) * 4);
            int dv[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    dv[i][c] = (q_ok && cur.y0 + py0 + i < a.H && cur.x0 + px0 + c < a.W) ? obase + (int)((i * osy + c * osx) * 4) : OOB;
            i32x4_t e_old[(OLDF || OLDA) ? 8 : 1], e_add[ADD ? 8 : 1], e_mask[MASK ? 8 : 1];
            if (OLDF || OLDA) {
#pragma unroll
                for (int i = 0; i < 8; ++i) e_old[i] = __builtin_amdgcn_raw_buffer_load_b128(ro, dv[i >> 1][i & 1], 0, 0);
            }
            if (ADD) {
                const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc(
                    reinterpret_cast<char*>(a.add.p) + ((size_t)cur.n * a.add.nstride + pb_) * 4, 0, 0x7fffff00, RSRC3);
#pragma unroll
                for (int i = 0; i < 8; ++i) e_add[i] = __builtin_amdgcn_raw_buffer_load_b128(ra_, dv[i >> 1][i & 1], 0, 0);
            }
            if (MASK) {
                const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(
                    reinterpret_cast<char*>(a.mask.p) + ((size_t)cur.n * a.mask.nstride + pb_) * 4, 0, 0x7fffff00, RSRC3);
#pragma unroll
                for (int i = 0; i < 8; ++i) e_mask[i] = __builtin_amdgcn_raw_buffer_load_b128(rm, dv[i >> 1][i & 1], 0, 0);
            }
            // columns 2 jh, 2 jh + 1 of M A: (M0 + s + t, d + 2 e) or (s + 4 t, d + 8 e + M5), s, d = M1 +- M2, t, e = M3 +- M4
            const float k1 = jh ? 4.f : 1.f, k2 = jh ? 8.f : 2.f, w0 = jh ? 0.f : 1.f, w1 = jh ? 1.f : 0.f;
            const f32x2 k1v = {k1, k1}, k2v = {k2, k2}, w0v = {w0, w0}, w1v = {w1, w1};
            const float* const pp = Pb + tile * PP + 4 * quad;
            const int nux = jh ? 5 : 0;
            f32x4 RA[6], RB[6];
#pragma unroll
            for (int xi = 0; xi < 6; ++xi) {
                const float* const pr = pp + (xi * 6) * 16 * PP;
                const f32x4 m1 = *reinterpret_cast<const f32x4*>(pr + 1 * 16 * PP), m2 = *reinterpret_cast<const f32x4*>(pr + 2 * 16 * PP);
                const f32x4 m3 = *reinterpret_cast<const f32x4*>(pr + 3 * 16 * PP), m4 = *reinterpret_cast<const f32x4*>(pr + 4 * 16 * PP);
                const f32x4 mx = *reinterpret_cast<const f32x4*>(pr + nux * 16 * PP);
                const f32x4 s = wino::add4(m1, m2), t = wino::add4(m3, m4);
                const f32x4 d = wino::subp4(pkc, m1, m2), ee = wino::subp4(pkc, m3, m4);
                RA[xi] = fma4v(mx, w0v, fma4v(t, k1v, s));
                RB[xi] = fma4v(mx, w1v, fma4v(ee, k2v, d));
            }
            WT(5);
            if (KQ - 1 >= GM::NBUF && has_next) {
                // chunk 2 of the next tile group goes where the products were: every thread has read them
                lds_barrier();
                stage_issue(nxt, KQ - 1);
            }
            WT(6);
            f32x4 bias4 = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (want_bias) bias4 = *reinterpret_cast<const f32x4*>(a.bias + min(n0 + 4 * quad, bias_max));
            const f32x2 two = {2.f, 2.f}, four = {4.f, 4.f}, eight = {8.f, 8.f};
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const f32x4* R = c ? RB : RA;
                const f32x4 s = wino::add4(R[1], R[2]), t = wino::add4(R[3], R[4]);
                const f32x4 d = wino::subp4(pkc, R[1], R[2]), ee = wino::subp4(pkc, R[3], R[4]);
                f32x4 y[4];
                y[0] = wino::add4(wino::add4(R[0], s), t);
                y[1] = fma4v(ee, two, d);
                y[2] = fma4v(t, four, s);
                y[3] = wino::add4(fma4v(ee, eight, d), R[5]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x4 r = wino::add4(y[i], bias4);
                    if (OLDF) r = wino::add4(r, __builtin_bit_cast(f32x4, e_old[2 * i + c]));
                    if (ADD) r = wino::add4(r, __builtin_bit_cast(f32x4, e_add[2 * i + c]));
                    r[0] = fmaxf(r[0], floor_v); r[1] = fmaxf(r[1], floor_v); r[2] = fmaxf(r[2], floor_v); r[3] = fmaxf(r[3], floor_v);
                    if (MASK) {
                        const f32x4 m = __builtin_bit_cast(f32x4, e_mask[2 * i + c]);
                        r[0] = m[0] > 0.f ? r[0] : 0.f; r[1] = m[1] > 0.f ? r[1] : 0.f;
                        r[2] = m[2] > 0.f ? r[2] : 0.f; r[3] = m[3] > 0.f ? r[3] : 0.f;
                    }
                    if (OLDA) r = wino::add4(r, __builtin_bit_cast(f32x4, e_old[2 * i + c]));
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, r), ro, dv[i][c], 0, 0);
                }
            }
        }
        WT(7);
#ifdef WINO_TRACE
        tr[0] += 1ull << 48;
#endif
        if (!has_next) break;
        cur = nxt;
        tg = ntg;
        first_it = false;
    }
#ifdef WINO_TRACE
    if (wp.trace && lane == 0)
        for (int q = 0; q < 8; ++q) wp.trace[((size_t)blockIdx.x * 4 + wave) * 8 + q] = tr[q];
#endif
}

template <int KQ, int NT, int EPI>
void launch_one(hipStream_t s, WinoParams& wp, int SX) {
    typedef Geom<KQ, NT> GM;
    static std::once_flag once;
    std::call_once(once, [&]() {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino4_kernel<KQ, NT, EPI>),
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
    DL4DS_LAUNCH((conv_wino4_kernel<KQ, NT, EPI>), dim3(8 * SX), dim3(256), GM::LDS_BYTES, s, wp);
    HIP_CHECK(hipGetLastError());
#ifdef WINO_TRACE
    if (wp.trace) {
        ++trace_n;
        HIP_CHECK(hipStreamSynchronize(s));
        std::vector<unsigned long long> h((size_t)8 * SX * 32);
        HIP_CHECK(hipMemcpy(h.data(), trace_buf, h.size() * 8, hipMemcpyDeviceToHost));
        static const char* nm[8] = {"prologue", "wait+bar", "issue", "transform+mfma", "barK+issue+writeP+bar", "C-read+transform", "bar+issue2", "store"};
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
            fprintf(stderr, "wino4<%d,%d,%d> wave %d: %d workgroups, %.1f iterations each; cycles: prologue %.0f | per iteration", KQ, NT, EPI, wv,
                    nwg, its / nwg, sum[0] / nwg);
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
        default: throw Dl4dsError("conv_wino4: epilogue form " + std::to_string(epi) + " is not built");
    }
}

}  // namespace wino4

void launch_wino4_32(hipStream_t s, WinoParams& wp, int SX, int epi);
void launch_wino4_22(hipStream_t s, WinoParams& wp, int SX, int epi);
