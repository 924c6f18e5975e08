// 3x3 convolution (stride 1, SAME) of the 40 / 48-channel layers with its fp32 PRODUCTS on the 16-bit matrix pipe of gfx950 (round 6;
// the product path of the single-pass <= 48 x <= 48 channel layers, see conv2d_split_forward).
// dl4ds/models/blocks.py:210-230, 401-454 (ResidualBlock / the sub-pixel block's convolutions) -- forward, and the data gradient as the
// same convolution with the flipped / transposed filter.
//
// Arithmetic: every operand is split exactly into three bf16 parts, x = xh + xm + xl (round-to-nearest residual splits: 8 + 8 + 8
// mantissa bits), and a product keeps the six terms above 2^-24 of it,
//     x w  ~  xh wh + xm wm + xh wm + xm wh + xh wl + xl wh          (dropped: xm wl, xl wm, xl wl  <=  3 * 2^-24 |x w|),
// accumulated in fp32 by v_mfma_f32_16x16x32_bf16.  Measured on a 16 x 16 tile with K = 288 / 1152 (profiles/mfma_split_r05.txt): error
// 3.9e-7 / 6.0e-7 of the result against 3.6e-7 / 9.1e-7 of v_mfma_f32_16x16x4_f32 -- fp32 arithmetic to rounding, not a narrower type.
// The 32 k-slots of one instruction carry TWO parts of 16 channels each, so the six terms of a 16-channel chunk take three instructions:
//     A = [wh | wm], B = [xh | xm]  ->  xh wh + xm wm          A = [wh | wm], B = [xm | xh]  ->  xm wh + xh wm
//     A = [wl | wh], B = [xh | xl]  ->  xh wl + xl wh
// (two filter forms held in registers, three forms of the staged pixels -- the same 16 bytes of LDS read with the lanes' halves swapped)
// i.e. 12 bf16 multiply-adds per fp32 multiply-add, on a pipe that issues 16 x the fp32 MFMA's rate: 9 taps x 48 x 48 channels cost
// 0.84 of the matrix cycles of the Winograd F(2x2,3x3) form on the fp32 pipe, with no transforms (conv_wino_kernel.h).
//
// Kernel: one workgroup of nine waves walks a strip of 32 output pixels x R rows.  Wave (kx, ct) holds the filter fragments of tap column
// kx for 16 output channels -- three tap rows x three 16-channel chunks x two operand forms = 72 registers -- for the whole launch.
// Per INPUT row: every thread splits one float4 of the row into the three bf16 planes in LDS ([pixel][part][48 channels], 288 bytes per
// pixel: the 16-byte reads of the MFMA's second operand are conflict-free at that pitch); after one barrier every wave reads its three
// operand forms of a 16-pixel group and chunk once (three ds_read_b128) and issues nine MFMAs -- the three tap rows feed three ROLLING
// accumulators (output rows y + 1, y, y - 1), so an input row is read from LDS once per tap column, not once per tap.  The finished row's
// three per-tap-column partial sums meet in LDS; bias / residual / ReLU mask / accumulation and the 16-byte stores are done by all
// threads after the row's barrier (one barrier per row, row buffers and partial sums double-buffered).
#include "ops.h"
#include "prof.h"
#include "launch.h"
#include <algorithm>
#include <cstdlib>
#include <string>
#include <vector>
#include <cstdio>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int OOB = (int)0x7fffff80, RSRC3 = 0x00020000;

constexpr int SPX = 32;                 // output pixels per strip row
constexpr int HPX = SPX + 2;            // staged pixels per row
constexpr int PIXB = 288;               // bytes per staged pixel: three parts x 48 channels x 2
constexpr int ROWB = HPX * PIXB;        // 9792
constexpr int RPIX = 52;                // floats per pixel of the partial sums: 48 + 4 -- a 16-byte store is serviced in groups of eight consecutive lanes
                                        // (eight pixels here): at 48 floats apart they hit the same banks four at a time (SQ_LDS_BANK_CONFLICT 0.41 of
                                        // the LDS cycles), at 52 the eight quads cover the 32 banks once
constexpr int REDF = 5 * SPX * RPIX;    // floats of one row's partial sums: [slot: the tile's four matrix waves, its helper wave][pixel][RPIX]
constexpr int NMW = 12, NHW = 4;        // matrix waves, helper waves
constexpr int NTHREADS = 64 * (NMW + NHW);
constexpr size_t LDS_BYTES = 3 * ROWB + 2 * REDF * sizeof(float) + 48 * sizeof(float);

struct SplitParams {
    TView in, out, add, mask;
    const float* bias;
    const u32x4* frag;      // [chunk][wave 0..15][unit i 0..1][ky][form 0 / 1][64 lanes]: the filter fragments as the lanes hold them
    int cin0, Cin;          // this pass reads input channels [cin0, cin0 + 48) (zero beyond Cin)
    int Cout, H, W;
    int nchunk;             // output-channel chunks of 48
    int nsx, nsy, R;        // strips per row, segments per column, rows per segment
    int nitems;             // per chunk: N * nsy * nsx
    int relu, old;          // old: the stored value is added -- 1: the earlier input-channel passes' partial sum (before bias / mask), 2: gradient accumulation (after)
    int final_pass;         // bias / residual / mask / ReLU only then
    unsigned long long* trace;   // (development) [block][wave][4] cycle sums: MFMA phase, partial-sum write, barrier wait, rows
};

// Work of a 16-output-channel tile ct = 9 units (tap column kx, 16-channel chunk c; all three tap rows), 27 in all, 18 MFMAs per unit and
// row.  Twelve matrix waves, four per tile, take two units each (w = 4 ct + j: units 2 j, 2 j + 1); the ninth unit of tile ct goes to
// helper wave ct.  Waves go to SIMDs cyclically, so a SIMD gets three matrix waves of different tiles and one helper: 126, 126, 126, 108
// MFMAs per row (nine equal waves: 162 on the SIMD that holds three of them).
__host__ __device__ inline int split_unit(int w, int i) {       // -> unit u = 3 kx + c of wave w's i-th unit, or -1
    if (w < NMW) return i < 2 ? 2 * (w & 3) + i : -1;
    return (i == 0 && w - NMW < 3) ? 8 : -1;
}
__host__ __device__ inline int split_tile(int w) { return w < NMW ? w >> 2 : w - NMW; }

// filter -> per-lane fragments.  One thread per (chunk, wave (matrix and helper), i, ky, form, lane): eight bf16 values = part P(form, kq) of
// w[ky][kx][cin0 + 16 c + 8 (kq & 1) + j][48 chunk + 16 ct + lane % 16], j = 0..7, kq = lane / 16, (kx, c) = the wave's i-th unit.
__device__ __forceinline__ u32x4 split_filter_entry(const float* __restrict__ w, int CinTot, int cin0, int Cout, int e) {
    int i = e;
    const int lane = i & 63; i >>= 6;
    const int form = i % 2; i /= 2;
    const int ky = i % 3; i /= 3;
    const int ui = i % 2; i /= 2;
    const int wv = i % (NMW + NHW);
    const int chunk = i / (NMW + NHW);
    const int u = split_unit(wv, ui), kx = u / 3, c = u % 3, ct = split_tile(wv);
    const int kq = lane >> 4, co = chunk * 48 + ct * 16 + (lane & 15);
    // which part this lane's k-slots carry: form 0 [h | m], form 1 [l | h]
    const int part = (kq < 2) ? (form == 0 ? 0 : 2) : (form == 0 ? 1 : 0);
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ci = cin0 + 16 * c + 8 * (kq & 1) + j;
        const float x = (u >= 0 && ci < CinTot && co < Cout) ? w[((size_t)(ky * 3 + kx) * CinTot + ci) * Cout + co] : 0.f;
        const __bf16 h = (__bf16)x;
        const float r1 = x - (float)h;
        const __bf16 m = (__bf16)r1;
        const __bf16 l = (__bf16)(r1 - (float)m);
        v[j] = part == 0 ? h : (part == 1 ? m : l);
    }
    u32x4 o;
    __builtin_memcpy(&o, &v, 16);
    return o;
}

__global__ void __launch_bounds__(256) split_filter_kernel(const float* __restrict__ w, u32x4* __restrict__ frag, int CinTot, int cin0,
                                                           int Cout, int nchunk, int total) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    (void)nchunk;
    frag[e] = split_filter_entry(w, CinTot, cin0, Cout, e);
}

// The fragments of ALL the layers a graph pass will run on this kernel, in one launch (the Winograd layers' scheme, conv_wino.hip: "transformed
// filters of a GRAPH's layers"): 8 launches of split_filter_kernel per cfg2 step become 2.  Job = (filter, its passes of 48 input channels).
struct SplitFilterJob { const float* w; u32x4* frag; int Cin, Cout, per_pass, total, first; };      // total: entries of all passes; first: the job's first block
constexpr int SPLIT_JOBS_MAX = 24;
struct SplitFilterJobs { SplitFilterJob j[SPLIT_JOBS_MAX]; int n, total; };                        // total: blocks
__global__ void __launch_bounds__(256) split_filter_batched_kernel(const SplitFilterJobs jobs) {
    int k = 0;                                                       // block -> job: wave-uniform (scalar loads from the kernel arguments)
#pragma unroll 1
    while (k + 1 < jobs.n && (int)blockIdx.x >= jobs.j[k + 1].first) ++k;
    const int e_all = ((int)blockIdx.x - jobs.j[k].first) * 256 + (int)threadIdx.x;
    if (e_all >= jobs.j[k].total) return;
    const int per_pass = jobs.j[k].per_pass, pass = e_all / per_pass;
    jobs.j[k].frag[e_all] = split_filter_entry(jobs.j[k].w, jobs.j[k].Cin, pass * 48, jobs.j[k].Cout, e_all - pass * per_pass);
}

// x = h + m + l exactly to 2^-24 |x|: h, m the upper halves of x and of the (exact) residual, l the residual's residual -- truncating
// splits (an AND and a subtraction per part; the parts have the sign of x), packed two to a dword by v_perm_b32
__device__ __forceinline__ void split4(const float4 x, bf16x4& h, bf16x4& m, bf16x4& l) {
    const float xs[4] = {x.x, x.y, x.z, x.w};
    unsigned hb[4], mb[4], lb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        hb[j] = __float_as_uint(xs[j]) & 0xffff0000u;
        const float r1 = xs[j] - __uint_as_float(hb[j]);
        mb[j] = __float_as_uint(r1) & 0xffff0000u;
        lb[j] = __float_as_uint(r1 - __uint_as_float(mb[j]));
    }
    unsigned hp[2], mp[2], lp[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        hp[j] = __builtin_amdgcn_perm(hb[2 * j + 1], hb[2 * j], 0x07060302u);      // (upper half of the odd element, upper half of the even one)
        mp[j] = __builtin_amdgcn_perm(mb[2 * j + 1], mb[2 * j], 0x07060302u);
        lp[j] = __builtin_amdgcn_perm(lb[2 * j + 1], lb[2 * j], 0x07060302u);
    }
    __builtin_memcpy(&h, hp, 8); __builtin_memcpy(&m, mp, 8); __builtin_memcpy(&l, lp, 8);
}

// the nine MFMAs of one (unit, 16-pixel group): three tap rows x the three operand pairings
__device__ __forceinline__ void split_mma9(const bf16x8 (&A)[3][2], const bf16x8 (&b)[3], f32x4& N, f32x4& M, f32x4& O) {
    N = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0][1], b[2], N, 0, 0, 0);
    M = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[1][1], b[2], M, 0, 0, 0);
    O = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[2][1], b[2], O, 0, 0, 0);
    N = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0][0], b[1], N, 0, 0, 0);
    M = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[1][0], b[1], M, 0, 0, 0);
    O = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[2][0], b[1], O, 0, 0, 0);
    N = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0][0], b[0], N, 0, 0, 0);
    M = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[1][0], b[0], M, 0, 0, 0);
    O = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[2][0], b[0], O, 0, 0, 0);
}
// the three forms of the second operand at byte offset o of the staged row: parts (h, h, m, m), (m, m, h, h), (h, h, l, l) by k-slot kq
__device__ __forceinline__ void split_fetch(const unsigned char* rb, int b1, int b2, int b3, int o, bf16x8 (&b)[3]) {
    b[2] = *reinterpret_cast<const bf16x8*>(rb + b3 + o);
    b[1] = *reinterpret_cast<const bf16x8*>(rb + b2 + o);
    b[0] = *reinterpret_cast<const bf16x8*>(rb + b1 + o);
}

#ifdef SPLIT_TRACE
// (development) cycle sums kept in registers, written once per wave at the end: work, barrier wait, rows
#define SPLIT_STAMP(k) do { const unsigned _n = (unsigned)__builtin_readcyclecounter(); _tw[(k) - 1] += _n - _t0; _t0 = _n; if ((k) == 2) ++_tw[2]; } while (0)
#define SPLIT_TRACE_DECL unsigned _tw[3] = {0, 0, 0}; unsigned _t0 = 0
#define SPLIT_TRACE_START _t0 = (unsigned)__builtin_readcyclecounter()
#define SPLIT_TRACE_END do { if (p.trace && lane == 0) { unsigned long long* tr = p.trace + ((size_t)blockIdx.x * (NMW + NHW) + wave) * 4; tr[0] = _tw[0]; tr[1] = _tw[1]; tr[3] = _tw[2]; } } while (0)
#else
#define SPLIT_STAMP(k) do { } while (0)
#define SPLIT_TRACE_DECL
#define SPLIT_TRACE_START
#define SPLIT_TRACE_END
#endif
// OLD / ADD / MASK: which epilogue operands exist (compiled apart: they stay in registers for a row)
template <bool OLD, bool ADD, bool MASK>
__global__ void __launch_bounds__(NTHREADS) conv_split_kernel(const SplitParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* rowbuf = smem;                                           // [3][HPX][PIXB]: a ring -- the row in use, the next one (complete), the one being staged
    float* red = reinterpret_cast<float*>(smem + 3 * ROWB);                 // [2][5][SPX][48]
    float* lbias = red + 2 * REDF;                                          // [48]: this chunk's bias (zero when absent / not the last pass)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = blockIdx.x % p.nchunk;
    const int slot = blockIdx.x / p.nchunk, nslots = gridDim.x / p.nchunk;
    const int n16 = lane & 15, kq = lane >> 4;
    // second operand: pixel (16 pt + n16 + kx) of the staged row; the unit's (kx, c) is the same for the whole wave: a scalar offset
    const int boff1 = n16 * PIXB + (kq >> 1) * 96 + (kq & 1) * 16;
    const int boff2 = n16 * PIXB + ((kq >> 1) ^ 1) * 96 + (kq & 1) * 16;
    const int boff3 = n16 * PIXB + (kq >> 1) * 192 + (kq & 1) * 16;
    const int ct = split_tile(wave);                                        // (helper wave 3: no unit)
    const int rslot = wave < NMW ? (wave & 3) : 4;                          // this wave's place among the tile's partial sums
    const int rdoff = (rslot * SPX + n16) * RPIX + (ct < 3 ? ct : 0) * 16 + kq * 4;
    SPLIT_TRACE_DECL;
    if (tid < 48) lbias[tid] = (p.bias && p.final_pass && chunk * 48 + tid < p.Cout) ? p.bias[chunk * 48 + tid] : 0.f;     // (visible after the first barrier)

    if (wave < NMW) {
        // ------------------------------------------------------------------------------------------------ matrix waves
        bf16x8 A[2][3][2];                                                  // [unit][ky][form]
        {
            const u32x4* f = p.frag + ((size_t)(chunk * (NMW + NHW) + wave) * 12) * 64 + lane;
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const u32x4 v = f[(size_t)i * 64];
                __builtin_memcpy(&A[i / 6][(i / 2) % 3][i % 2], &v, 16);
            }
        }
        int uo[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int u = split_unit(wave, i);
            uo[i] = __builtin_amdgcn_readfirstlane((u / 3) * PIXB + (u % 3) * 32);
        }
        // epilogue: thread -> (pixel epx, output-channel quad eq) of the finished row (the first six matrix waves)
        const int epx = tid / 12, eq = tid - epx * 12;
        const int co = chunk * 48 + eq * 4;
        const bool e_el = tid < SPX * 12 && co < p.Cout;
        const int out_ys = (p.out.d2s > 1 ? p.out.d2s * p.out.d2s : 1) * p.W * p.out.ld * 4;       // bytes between rows y and y + 1
        const int add_ys = ADD ? (p.add.d2s > 1 ? p.add.d2s * p.add.d2s : 1) * p.W * p.add.ld * 4 : 0;
        const int mask_ys = MASK ? (p.mask.d2s > 1 ? p.mask.d2s * p.mask.d2s : 1) * p.W * p.mask.ld * 4 : 0;
        for (int item = slot; item < p.nitems; item += nslots) {
            int q = item;
            const int sx = q % p.nsx; q /= p.nsx;
            const int sy = q % p.nsy;
            const int n = q / p.nsy;
            const int y_lo = sy * p.R, y_hi = min(y_lo + p.R, p.H);
            // byte offsets inside image n (row 0); what lies outside the tensor gets an out-of-range offset, for which the buffer unit
            // returns zero (loads) / drops the access (no branch around a load: a load inside a conditional block ends the block with
            // s_waitcnt vmcnt(0))
            const int exg = sx * SPX + epx;
            const bool e_ok = e_el && exg < p.W;
            const int ooff = e_ok ? (int)(view_off(p.out, 0, 0, exg, co) * 4) : OOB;
            const int aoff = (ADD && e_ok) ? (int)(view_off(p.add, 0, 0, exg, co) * 4) : OOB;
            const int moff = (MASK && e_ok) ? (int)(view_off(p.mask, 0, 0, exg, co) * 4) : OOB;
            const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out.p + (size_t)n * p.out.nstride, 0, 0x7fffff00, RSRC3);
            const __amdgpu_buffer_rsrc_t rs_add = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(ADD ? p.add.p + (size_t)n * p.add.nstride : p.out.p), 0, 0x7fffff00, RSRC3);
            const __amdgpu_buffer_rsrc_t rs_mask = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(MASK ? p.mask.p + (size_t)n * p.mask.nstride : p.out.p), 0, 0x7fffff00, RSRC3);
            i32x4 ov = {0, 0, 0, 0}, av = {0, 0, 0, 0}, mv = {0, 0, 0, 0};
            // the epilogue's operands for output row yo, requested one row ahead of their use
            auto request = [&](int yo) __attribute__((always_inline)) {
                const int eoff = (yo >= y_lo && yo < y_hi && ooff != OOB) ? 0 : OOB;     // (one select; the operands' offsets are added to it)
                if constexpr (OLD) ov = __builtin_amdgcn_raw_buffer_load_b128(rs_out, (eoff | ooff) + (eoff == 0 ? yo * out_ys : 0), 0, 0);
                if constexpr (ADD) av = __builtin_amdgcn_raw_buffer_load_b128(rs_add, (eoff | aoff) + (eoff == 0 ? yo * add_ys : 0), 0, 0);
                if constexpr (MASK) mv = __builtin_amdgcn_raw_buffer_load_b128(rs_mask, (eoff | moff) + (eoff == 0 ? yo * mask_ys : 0), 0, 0);
            };
            // output row yo from the five per-wave partial sums of every channel tile in red[buf]
            auto finish = [&](int yo, int buf) __attribute__((always_inline)) {
                const int eoff = (yo >= y_lo && yo < y_hi && ooff != OOB) ? 0 : OOB;
                const float* r0 = red + buf * REDF + epx * RPIX + eq * 4;
                const float4 a0 = *reinterpret_cast<const float4*>(r0);
                const float4 a1 = *reinterpret_cast<const float4*>(r0 + SPX * RPIX);
                const float4 a2 = *reinterpret_cast<const float4*>(r0 + 2 * SPX * RPIX);
                const float4 a3 = *reinterpret_cast<const float4*>(r0 + 3 * SPX * RPIX);
                const float4 a4 = *reinterpret_cast<const float4*>(r0 + 4 * SPX * RPIX);
                float4 v = make_float4(((a0.x + a1.x) + (a2.x + a3.x)) + a4.x, ((a0.y + a1.y) + (a2.y + a3.y)) + a4.y,
                                       ((a0.z + a1.z) + (a2.z + a3.z)) + a4.z, ((a0.w + a1.w) + (a2.w + a3.w)) + a4.w);
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f), r = o, mk = o;
                if constexpr (OLD) __builtin_memcpy(&o, &ov, 16);
                if constexpr (ADD) __builtin_memcpy(&r, &av, 16);
                if constexpr (MASK) __builtin_memcpy(&mk, &mv, 16);
                if (OLD && p.old == 1) { v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }      // the earlier input-channel passes' partial sum
                if (p.final_pass) {
                    const float4 b4 = *reinterpret_cast<const float4*>(lbias + eq * 4);
                    v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
                    if (ADD) { v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
                    if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    if (MASK) { v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f; v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f; }
                    if (OLD && p.old == 2) { v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }  // gradient accumulation: out += masked result
                }
                i32x4 vi;
                __builtin_memcpy(&vi, &v, 16);
                __builtin_amdgcn_raw_buffer_store_b128(vi, rs_out, (eoff | ooff) + (eoff == 0 ? yo * out_ys : 0), 0, 0);
            };
            f32x4 acc[3][2];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) acc[a][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            int y = y_lo - 1;
            __syncthreads();                                                // (the helpers have staged the first row)
            SPLIT_TRACE_START;
            // One input row y: N takes tap row 0 (output row y + 1), M tap row 1 (row y), O tap row 2 (row y - 1, finished here).  Four
            // (unit, pixel group) steps, the operand forms of step i + 1 requested before the MFMAs of step i.
            // (the first step's operand forms are requested before the previous row's barrier: the row after the one in use is complete)
            bf16x8 bA[3], bB[3];
            split_fetch(rowbuf, boff1, boff2, boff3, uo[0], bA);
            auto row = [&](int K, f32x4 (&N)[2], f32x4 (&M)[2], f32x4 (&O)[2]) __attribute__((always_inline)) {
                const unsigned char* rb = rowbuf + K * ROWB;
                __builtin_amdgcn_sched_barrier(0);
                split_fetch(rb, boff1, boff2, boff3, uo[0] + 16 * PIXB, bB);
                __builtin_amdgcn_sched_barrier(0);
                split_mma9(A[0], bA, N[0], M[0], O[0]);
                __builtin_amdgcn_sched_barrier(0);
                if (wave < 6) {                       // (under the other waves' MFMAs) output row y - 2, then the next one's operands
                    finish(y - 2, (y - 1) & 1);
                    request(y - 1);
                }
                __builtin_amdgcn_sched_barrier(0);
                split_fetch(rb, boff1, boff2, boff3, uo[1], bA);
                __builtin_amdgcn_sched_barrier(0);
                split_mma9(A[0], bB, N[1], M[1], O[1]);
                __builtin_amdgcn_sched_barrier(0);
                split_fetch(rb, boff1, boff2, boff3, uo[1] + 16 * PIXB, bB);
                __builtin_amdgcn_sched_barrier(0);
                split_mma9(A[1], bA, N[0], M[0], O[0]);
                __builtin_amdgcn_sched_barrier(0);
                split_fetch(rowbuf + ((K + 1) % 3) * ROWB, boff1, boff2, boff3, uo[0], bA);           // the next row's first step
                __builtin_amdgcn_sched_barrier(0);
                split_mma9(A[1], bB, N[1], M[1], O[1]);
                __builtin_amdgcn_sched_barrier(0);
                // the finished row's partial sum over this wave's units: lane holds output channels 16 ct + 4 kq .. + 3 of pixel 16 pt + n16
                float* rd = red + (y & 1) * REDF + rdoff;
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) {
                    *reinterpret_cast<f32x4*>(rd + pt * 16 * RPIX) = O[pt];
                    O[pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                SPLIT_STAMP(1);
                __syncthreads();
                SPLIT_STAMP(2);
                ++y;
            };
            // rows y_lo - 1 .. y_hi: three at a time so that the rolling accumulators (and the ring's buffers) keep their names
            while (y <= y_hi) {
                row(0, acc[0], acc[1], acc[2]);
                if (y > y_hi) break;
                row(1, acc[2], acc[0], acc[1]);
                if (y > y_hi) break;
                row(2, acc[1], acc[2], acc[0]);
            }
            if (wave < 6) finish(y - 2, (y - 1) & 1);             // the last output row (y = y_hi + 1 here)
        }
        SPLIT_TRACE_END;
        return;
    }

    // ---------------------------------------------------------------------------------------------------- helper waves
    // Staging: per input row y, while the matrix waves work on it, input row y + 2 (loaded during the previous row) is split into the
    // ring buffer the previous row's MFMAs have left and the loads of row y + 3 are issued.  Helper wave h < 3 also takes the ninth unit of
    // channel tile h.  (The epilogue of output row y - 2 -- its partial sums were completed by the previous row's barrier -- is done by the
    // first six matrix waves between two of their MFMA groups.)
    const int hid = tid - 64 * NMW;                                         // 0 .. 255
    constexpr int NH = 64 * NHW;
    const bool hunit = ct < 3;
    bf16x8 A[3][2];
    {
        const u32x4* f = p.frag + ((size_t)(chunk * (NMW + NHW) + wave) * 12) * 64 + lane;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const u32x4 v = f[(size_t)i * 64];
            __builtin_memcpy(&A[i / 2][i % 2], &v, 16);
        }
    }
    const int uo = 2 * PIXB + 2 * 32;                                       // unit 8 = (kx 2, chunk 2)
    const int in_ys = (p.in.d2s > 1 ? p.in.d2s * p.in.d2s : 1) * p.W * p.in.ld * 4;          // bytes between rows y and y + 1
    // staging elements (pixel spx of the staged row, channel quad sq): two per thread
    int spx[2], sq[2];
    bool s_el[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int e = hid + k * NH;
        spx[k] = e / 12; sq[k] = e - spx[k] * 12; s_el[k] = e < HPX * 12;
    }
    for (int item = slot; item < p.nitems; item += nslots) {
        int q = item;
        const int sx = q % p.nsx; q /= p.nsx;
        const int sy = q % p.nsy;
        const int n = q / p.nsy;
        const int x0 = sx * SPX, y_lo = sy * p.R, y_hi = min(y_lo + p.R, p.H);
        // per-thread byte offsets inside image n (row 0); everything outside the tensor gets an out-of-range offset, for which the
        // buffer unit returns zero (loads) / drops the access (no branch around a load: a load inside a conditional block ends the
        // block with s_waitcnt vmcnt(0))
        int soff[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int sxg = x0 - 1 + spx[k], sc = p.cin0 + sq[k] * 4;
            const bool s_ok = s_el[k] && sxg >= 0 && sxg < p.W && sc < p.Cin;
            soff[k] = s_ok ? (int)(view_off(p.in, 0, 0, sxg, sc) * 4) : OOB;
        }
        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(p.in.p + (size_t)n * p.in.nstride, 0, 0x7fffff00, RSRC3);

        i32x4 nxt[2];
        auto load_row = [&](int y) __attribute__((always_inline)) {
            const bool ok = y >= 0 && y < p.H;
#pragma unroll
            for (int k = 0; k < 2; ++k) nxt[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (ok && soff[k] != OOB) ? soff[k] + y * in_ys : OOB, 0, 0);
        };
        auto stage_row = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (s_el[k]) {
                    float4 v;
                    __builtin_memcpy(&v, &nxt[k], 16);
                    bf16x4 h, m, l;
                    split4(v, h, m, l);
                    unsigned char* d = rowbuf + buf * ROWB + spx[k] * PIXB + sq[k] * 8;
                    *reinterpret_cast<bf16x4*>(d) = h;
                    *reinterpret_cast<bf16x4*>(d + 96) = m;
                    *reinterpret_cast<bf16x4*>(d + 192) = l;
                }
            }
        };
        f32x4 acc[3][2];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) acc[a][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // prologue: input rows y_lo - 1 and y_lo staged, row y_lo + 1 in registers
        int y = y_lo - 1;
        load_row(y);
        stage_row(0);
        load_row(y + 1);
        stage_row(1);
        load_row(y + 2);
        __syncthreads();
        SPLIT_TRACE_START;
        auto row = [&](int K, f32x4 (&N)[2], f32x4 (&M)[2], f32x4 (&O)[2]) __attribute__((always_inline)) {
            if (hunit) {                              // (first: the matrix pipe is what the row waits for)
                const unsigned char* rb = rowbuf + K * ROWB;
                bf16x8 bA[3];
                __builtin_amdgcn_s_setprio(3);                  // (the youngest wave of its SIMD: without this its two MFMA groups wait behind the matrix waves' -- ten calls 3.27 -> 3.19 ms)
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) {
                    __builtin_amdgcn_sched_barrier(0);
                    split_fetch(rb, boff1, boff2, boff3, uo + pt * 16 * PIXB, bA);
                    split_mma9(A, bA, N[pt], M[pt], O[pt]);
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(0);
                float* rd = red + (y & 1) * REDF + rdoff;
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) {
                    *reinterpret_cast<f32x4*>(rd + pt * 16 * RPIX) = O[pt];
                    O[pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            stage_row((K + 2) % 3);                   // input row y + 2 (loaded during the previous row)
            load_row(y + 3);
            SPLIT_STAMP(1);
            __syncthreads();
            SPLIT_STAMP(2);
            ++y;
        };
        while (y <= y_hi) {
            row(0, acc[0], acc[1], acc[2]);
            if (y > y_hi) break;
            row(1, acc[2], acc[0], acc[1]);
            if (y > y_hi) break;
            row(2, acc[1], acc[2], acc[0]);
        }
    }
    SPLIT_TRACE_END;
}

template <bool OLD, bool ADD, bool MASK>
void launch_one(hipStream_t s, const SplitParams& p, int grid) {
    static bool once = false;
    if (!once) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_split_kernel<OLD, ADD, MASK>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)LDS_BYTES));
        once = true;
    }
    DL4DS_LAUNCH((conv_split_kernel<OLD, ADD, MASK>), dim3(grid), dim3(NTHREADS), LDS_BYTES, s, p);
    HIP_CHECK(hipGetLastError());
}
void launch_form(hipStream_t s, const SplitParams& p, int grid) {
    const int f = (p.old ? 4 : 0) | (p.add.p ? 2 : 0) | (p.mask.p ? 1 : 0);
    switch (f) {
        case 0: launch_one<false, false, false>(s, p, grid); break;
        case 1: launch_one<false, false, true>(s, p, grid); break;
        case 2: launch_one<false, true, false>(s, p, grid); break;
        case 3: launch_one<false, true, true>(s, p, grid); break;
        case 4: launch_one<true, false, false>(s, p, grid); break;
        case 5: launch_one<true, false, true>(s, p, grid); break;
        case 6: launch_one<true, true, false>(s, p, grid); break;
        default: launch_one<true, true, true>(s, p, grid); break;
    }
}

int cu_count() {
    static const int n = [] {
        int dev = 0, v = 0;
        HIP_CHECK(hipGetDevice(&dev));
        HIP_CHECK(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
        return v;
    }();
    return n;
}

// The filter fragments of a call live in a scratch buffer of the call's STREAM (one per stream, like the Winograd kernels' scratch: two
// streams may each have a convolution in flight; on one stream the fragment kernel and the convolution that reads it are ordered).  They are
// rebuilt on every call -- one short launch.
float* frag_scratch(hipStream_t s, size_t floats) {
    struct Slot { hipStream_t s; float* buf; size_t cap; };
    static std::vector<Slot> slots;
    for (auto& e : slots) {
        if (e.s != s) continue;
        if (e.cap < floats) {
            HIP_CHECK(hipStreamSynchronize(s));
            HIP_CHECK(hipFree(e.buf));
            HIP_CHECK(hipMalloc((void**)&e.buf, floats * sizeof(float)));
            e.cap = floats;
        }
        return e.buf;
    }
    Slot e{s, nullptr, std::max<size_t>(floats, 1 << 18)};
    HIP_CHECK(hipMalloc((void**)&e.buf, e.cap * sizeof(float)));
    slots.push_back(e);
    return e.buf;
}

// Inside a graph pass a layer's fragments have a buffer of their own and are rebuilt by ONE launch per pass for all registered layers
// (split_filters_refresh, called where the Winograd filters are refreshed); freshness never outlives a forward pass.  Outside a pass (the
// op-level API) nothing is registered or trusted: the stream's scratch and one launch per call.
struct SplitFilterEntry { const float* w; int Cin, Cout, per_pass, passes, kind; hipStream_t stream; u32x4* frag; bool fresh; };
std::vector<SplitFilterEntry>& split_entries() { static std::vector<SplitFilterEntry> v; return v; }
constexpr size_t SPLIT_ENTRIES_MAX = 512;

u32x4* split_filter_lookup(hipStream_t s, const float* w, int Cin, int Cout, int per_pass, int passes, bool& need) {
    need = true;
    int kind = 0;
    if (!wino_pass_active(kind)) return reinterpret_cast<u32x4*>(frag_scratch(s, (size_t)per_pass * passes * 4));
    auto& es = split_entries();
    for (auto& e : es)
        if (e.w == w && e.Cin == Cin && e.Cout == Cout && e.per_pass == per_pass && e.passes == passes && e.stream == s) {
            need = !e.fresh;
            e.fresh = true;                      // (the caller builds them now if they were not)
            return e.frag;
        }
    if (es.size() >= SPLIT_ENTRIES_MAX) return reinterpret_cast<u32x4*>(frag_scratch(s, (size_t)per_pass * passes * 4));
    SplitFilterEntry e{w, Cin, Cout, per_pass, passes, kind, s, nullptr, true};
    HIP_CHECK(hipMalloc((void**)&e.frag, (size_t)per_pass * passes * sizeof(u32x4)));
    es.push_back(e);
    return e.frag;
}

}  // namespace

void split_filters_invalidate(const float* lo, const float* hi) {
    for (auto& e : split_entries())
        if (e.w >= lo && e.w < hi) e.fresh = false;
}

void split_filters_release(const float* lo, const float* hi) {
    auto& es = split_entries();
    for (size_t i = 0; i < es.size();) {
        if (es[i].w >= lo && es[i].w < hi) { (void)hipFree(es[i].frag); es[i] = es.back(); es.pop_back(); }
        else ++i;
    }
}

void split_filters_refresh(hipStream_t s, const float* lo, const float* hi, int kind) {
    SplitFilterJobs jobs;
    jobs.n = 0; jobs.total = 0;
    auto flush = [&]() {
        if (!jobs.n) return;
        ProfScope ps(s, "split_filters", 0.0, 16.0 * jobs.total * 256);
        DL4DS_LAUNCH(split_filter_batched_kernel, dim3(jobs.total), dim3(256), 0, s, jobs);
        HIP_CHECK(hipGetLastError());
        jobs.n = 0; jobs.total = 0;
    };
    for (auto& e : split_entries()) {
        if (e.fresh || e.kind != kind || e.stream != s || e.w < lo || e.w >= hi) continue;
        if (jobs.n == SPLIT_JOBS_MAX) flush();
        jobs.j[jobs.n++] = SplitFilterJob{e.w, e.frag, e.Cin, e.Cout, e.per_pass, e.per_pass * e.passes, jobs.total};
        jobs.total += cdiv(e.per_pass * e.passes, 256);
        e.fresh = true;
    }
    flush();
}

bool conv2d_split_forward(hipStream_t s, const TView& in, const float* w, const TView& out, const ConvEpilogue& ep) {
    // Measured (round 6, profiles/conv_split_r06.txt): 10 % faster than the Winograd F(2x2,3x3) kernels on the fp32 pipe for the layers with
    // ONE pass of <= 48 input channels and <= 48 output channels (0.199 vs 0.21-0.23 ms at 64 x 128^2), slower for 48 -> 192 / 192 -> 48
    // (0.76 / 0.83 vs 0.74 / 0.84).  So: ON by default for the former (cfg2: eight of the ten <3,3> layers, 5 075 -> 5 175 samples/s in
    // three alternating runs), the latter only with DL4DS_SPLIT=1 (all eligible shapes: an experiment switch); DL4DS_NO_SPLIT=1 turns the
    // kernel off altogether -- the A/B switch of the product, and what `arith` in the bench line reports against.
    const char* force = test_env("DL4DS_SPLIT_FORCE");           // (tests: small grids too; "<k>": k workgroups per output-channel chunk)
    if (getenv("DL4DS_NO_SPLIT")) return false;
    const bool every_shape = getenv("DL4DS_SPLIT") != nullptr || force != nullptr;
    if (in.sc || ep.pool) return false;
    if (!in.vec || !out.vec || (in.C & 3) || (out.C & 3) || (ep.add.p && !ep.add.vec) || (ep.mask.p && !ep.mask.vec)) return false;
    if ((((uintptr_t)ep.bias) & 15) != 0) return false;
    // the layers the <3,3> Winograd form takes: 33 .. 48 input channels (or passes of 48), output channels in chunks of 48
    if (in.C <= 32 || (in.C > 48 && in.C % 48 != 0)) return false;
    if (out.C <= 32 || cdiv(out.C, 32) * 32 < cdiv(out.C, 48) * 48) return false;
    const int passes = cdiv(in.C, 48);
    if (!every_shape && (passes > 1 || out.C > 48)) return false;
    if (passes > 1 && ep.accumulate) return false;
    if (ep.add.p && (ep.add.C != out.C)) return false;
    // (the kernel addresses an image with 32-bit byte offsets from the image's base)
    auto span = [](const TView& v) { const size_t r = std::max(v.d2s, 1); return (size_t)(v.H + 2) * v.W * r * r * v.ld * 4; };
    if (span(in) >= (1ull << 31) || span(out) >= (1ull << 31) || (ep.add.p && span(ep.add) >= (1ull << 31)) ||
        (ep.mask.p && span(ep.mask) >= (1ull << 31))) return false;
    SplitParams p;
    p.in = in; p.out = out; p.add = ep.add; p.mask = ep.mask; p.bias = ep.bias;
    p.Cin = in.C; p.Cout = out.C; p.H = in.H; p.W = in.W;
    p.nchunk = cdiv(out.C, 48);
    p.nsx = cdiv(in.W, SPX);
    p.R = in.H <= 32 ? in.H : 32;
    // strips of 16 rows when 32-row strips would leave CUs without one (cfg2 at per-GPU batch 8: 128 -> 256 strip segments, the eight layers
    // stay on this kernel: 2 986 -> 3 160 samples/s; with a strip per CU either way 32 rows win: 4 133 vs 4 014 at batch 16, 5 094 vs 4 988 at 64)
    if (in.H > 16 && (long)p.nsx * cdiv(in.H, p.R) * in.N * cdiv(out.C, 48) < (long)cu_count()) p.R = 16;
    if (const char* r = test_env("DL4DS_SPLIT_R")) p.R = std::max(1, std::min(atoi(r), in.H));
    p.nsy = cdiv(in.H, p.R);
    const long items = (long)p.nsx * p.nsy * in.N;
    // fewer strips than CUs: the Winograd / direct kernels.  (One strip segment per workgroup is enough: cfg2 at per-GPU batch 16 = 256 segments
    // gains 5 % on the step, 3 950 -> 4 160 samples/s; the first form asked for two per CU.)
    if (items >= (1l << 24) || (!force && items * p.nchunk < (long)cu_count())) return false;
    p.nitems = (int)items;
    p.relu = ep.relu;
    const double px = (double)in.N * in.H * in.W;
    const double fl = 2.0 * px * 9.0 * in.C * out.C;
    ProfScope ps(s, std::string("conv_split<3,3>") + (test_env("DL4DS_SPLIT_TAG_FORMS") ? std::string("f") + std::to_string((ep.accumulate ? 4 : 0) | (ep.add.p ? 2 : 0) | (ep.mask.p ? 1 : 0)) + "c" + std::to_string(in.C) + "_" + std::to_string(out.C) : std::string()), fl,
                 4.0 * (px * (in.C + out.C * (1 + (ep.add.p ? 1 : 0) + (ep.mask.p ? 1 : 0) + (ep.accumulate ? 1 : 0))) + 9.0 * in.C * out.C), fl);
    const int per_pass = p.nchunk * (NMW + NHW) * 12 * 64;                  // uint4 entries
    bool need_frag = true;
    u32x4* frag = split_filter_lookup(s, w, in.C, out.C, per_pass, passes, need_frag);
    int grid = std::min((int)std::min<long>(items, cu_count()), cu_count()) * 1;
    grid = std::max(1, grid / p.nchunk) * p.nchunk;
    if (grid > cu_count()) grid = (cu_count() / p.nchunk) * p.nchunk;
    if (force && atoi(force) > 0) grid = std::min(grid, atoi(force) * p.nchunk);
    p.trace = nullptr;
    static unsigned long long* trace_buf = nullptr;
    const bool tracing = test_env("DL4DS_SPLIT_TRACE") != nullptr;
    if (tracing) {
        if (!trace_buf) HIP_CHECK(hipMalloc((void**)&trace_buf, 256 * (NMW + NHW) * 4 * 8));
        HIP_CHECK(hipMemsetAsync(trace_buf, 0, 256 * (NMW + NHW) * 4 * 8, s));
        p.trace = trace_buf;
    }
    for (int ps_ = 0; ps_ < passes; ++ps_) {
        p.cin0 = ps_ * 48;
        p.frag = frag + (size_t)per_pass * ps_;
        if (need_frag) {
            DL4DS_LAUNCH(split_filter_kernel, dim3(cdiv(per_pass, 256)), dim3(256), 0, s, w, const_cast<u32x4*>(p.frag), in.C, p.cin0, out.C, p.nchunk,
                         per_pass);
            HIP_CHECK(hipGetLastError());
        }
        p.final_pass = ps_ == passes - 1;
        p.old = (ps_ > 0) ? 1 : (ep.accumulate ? 2 : 0);
        if (!p.final_pass) { p.add.p = nullptr; p.mask.p = nullptr; } else { p.add = ep.add; p.mask = ep.mask; }
        launch_form(s, p, grid);
    }
    if (tracing) {
        HIP_CHECK(hipStreamSynchronize(s));
        std::vector<unsigned long long> h((size_t)grid * (NMW + NHW) * 4);
        HIP_CHECK(hipMemcpy(h.data(), trace_buf, h.size() * 8, hipMemcpyDeviceToHost));
        for (int w = 0; w < NMW + NHW; ++w) {
            double a = 0, b = 0, c = 0, r = 0;
            for (int g = 0; g < grid; ++g) { a += h[((size_t)g * (NMW + NHW) + w) * 4]; b += h[((size_t)g * (NMW + NHW) + w) * 4 + 1]; c += h[((size_t)g * (NMW + NHW) + w) * 4 + 2]; r += h[((size_t)g * (NMW + NHW) + w) * 4 + 3]; }
            fprintf(stderr, "conv_split trace wave %d: per row: work %.0f, barrier wait %.0f cycles (%.0f rows per workgroup; Cin %d Cout %d passes %d)\n",
                    w, a / r, b / r, r / grid, in.C, out.C, passes); (void)c;
        }
    }
    return true;
}
