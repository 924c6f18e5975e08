// dl4ds_amd -- implicit-GEMM 3x3 / 1x1 convolution with the filter streamed from L2 straight into MFMA operands.
//
// The MFMA-bound layers of the residual backbone and of SubpixelConvolution (blocks.py:210-230, 433-454: 48->48,
// 48->192 and, as dgrad, 192->48 at 128^2 .. 256^2) spent ~45 % of their time outside the matrix cores in the
// LDS-staged kernels: every 1.9 us stage waited for its filter slice (global -> registers -> LDS -> barrier) and the
// eight waves of a CU marched through prologue / stages / epilogue in lock-step.  Here the filter never touches LDS:
//   * output channels are assigned to MFMA rows as cout = n0 + NT*row + j (j = accumulator tile), so ONE
//     global_load_dwordx{NT} per lane and k-step yields the first operand of all NT tiles; the rows of one k-step are
//     contiguous 64*NT-byte segments that stay hot in L2 (the whole filter is 83 KB for 48->48);
//   * the K axis is ordered (tap, e) with MFMA k-slot q owning cin group [E*q, E*q+E) (E = CK/4), so the second
//     operand (pixels) comes from the LDS halo tile as ds_read_b128/b64 at IMMEDIATE offsets, each feeding 4 (2) k-steps;
//   * there is no barrier inside the K loop; waves drift apart and one wave's loads/epilogue overlap other waves' MFMAs;
//     LDS holds only the input tile (67 KB at CK = 48), two workgroups per CU;
//   * with that row assignment a lane ends up with 4*NT CONSECUTIVE output channels of one pixel -> NT float4 stores,
//     also through depth_to_space views;
//   * workgroups are numbered so that each XCD walks a contiguous range of tiles and the n-blocks of one tile run
//     back to back on the same XCD (the input tile is fetched from HBM once per L2).
// Rows beyond Cin (padded K) read real filter rows of the next tap (finite values) against ZEROED pixels, so the
// inner loop needs no masks; only the very last tap clamps its row index.
#include "ops.h"
#include "prof.h"
#include "conv_kernels.h"
#include <algorithm>
#include <type_traits>
#include <mutex>
#include <vector>

namespace {

template <int NT> struct WVec;
template <> struct WVec<1> { typedef float T; };
template <> struct WVec<2> { typedef float T __attribute__((ext_vector_type(2), aligned(4))); };
template <> struct WVec<3> { typedef float T __attribute__((ext_vector_type(3), aligned(4))); };
template <> struct WVec<4> { typedef float T __attribute__((ext_vector_type(4), aligned(4))); };

template <int NT>
__device__ __forceinline__ float wget(const typename WVec<NT>::T& v, int j) {
    if constexpr (NT == 1) return v;
    else return v[j];
}

struct StreamParams {
    ConvParams c;
    int nblk;           // n-blocks (16*NT couts each)
    int ntiles;         // spatial tiles * batch
    int per_xcd;        // tiles per XCD (contiguous range)
    unsigned m_nblk;
    int cw;             // filter row stride = Cout, or Cout padded up to whole 16*NT blocks (zero columns) when Cout % NT != 0
#ifdef STREAM_TRACE
    unsigned long long* trace;   // diagnostics build: [workgroup][8] = s_memrealtime stamps at the phase boundaries + HW_ID
#endif
};

constexpr int kStreamThreads = 256;

// MT = accumulator rows (of 16 pixels) per wave; tile = 16 x (4*MT) pixels.  MT = 8 halves the filter traffic per MFMA
// (each streamed fragment now feeds 8 x NT tiles); its 16x32 halo tile only fits two-per-CU for channel chunks
// <= 24 (E <= 6), so 48-channel layers run as two chunks.  Measured on the 192->48 dgrad: 102 -> 115 TFLOP/s.
template <int KS, int E, int NT, int MT>
__global__ void __launch_bounds__(kStreamThreads, 2) conv_stream_kernel(const StreamParams sp) {
    const ConvParams& a = sp.c;
    constexpr int G = (E % 4 == 0) ? 4 : 2;          // k-steps fed by one LDS read
    constexpr int NGRP = E / G;
    constexpr int CK = 4 * E;                        // input channels per chunk
    constexpr int P = CK + G;                        // LDS pixel pitch: P/G odd -> conflict-free b128 / b64 reads
    constexpr int TW = 16, TH = 4 * MT, NTHR = kStreamThreads, PAD = KS / 2;
    constexpr int TWH = TW + KS - 1, THH = TH + KS - 1, HPIX = TWH * THH;
    constexpr int KK = KS * KS, NGS = KK * NGRP;     // group-steps per chunk
    constexpr int Q4 = CK / 4;
    constexpr int TOTAL = HPIX * Q4, ITERS = (TOTAL + NTHR - 1) / NTHR;
#ifndef STREAM_WPD
#define STREAM_WPD 2
#endif
    constexpr int WPD = STREAM_WPD;                  // filter prefetch distance in group-steps
    typedef typename WVec<NT>::T wvec_t;
    extern __shared__ __attribute__((aligned(16))) float tile[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;

    // ---- which (tile, n-block): XCD = id % 8 owns tiles [xcd*per_xcd, (xcd+1)*per_xcd), n-blocks innermost
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int tl = fast_div(slot, sp.m_nblk);
    const int nb = slot - tl * sp.nblk;
    const int t = xcd * sp.per_xcd + tl;
    if (tl >= sp.per_xcd || t >= sp.ntiles) return;
    const int q = fast_div(t, a.m_txy[0]);
    const int bx = t - q * a.tiles_x;
    const int n = fast_div(q, a.m_txy[1]);
    const int by = q - n * a.tiles_y;
    const int x0 = bx * TW, y0 = by * TH;
    const int n0 = nb * 16 * NT;

    // ---- filter addressing: lane (row l15, k-slot lq) reads floats [co, co+NT) of row (tap*Cin + c0 + E*lq + e)
    const int co_lane = min(n0 + NT * l15, sp.cw - NT);           // rows beyond Cout are never stored
    const int last_row = KK * a.Cin - 1;
    const float* wlane = a.w + co_lane;

    const float* rd = tile + ((wave * MT) * TWH + l15) * P + E * lq;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#ifdef STREAM_TRACE
    // phase timeline of every workgroup (tools/stream_trace.py): start, [tile staged, chunk's MFMAs done] per chunk, end
    int tr_k = 0;
    auto stamp = [&]() {
        if (tid == 0 && sp.trace && tr_k < 7) sp.trace[(size_t)blockIdx.x * 8 + tr_k++] = wall_clock64();
    };
    if (tid == 0 && sp.trace) sp.trace[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    stamp();
#define STAMP() stamp()
#else
#define STAMP()
#endif
    for (int c0 = 0; c0 < a.Cin; c0 += CK) {
        if (c0 > 0) __syncthreads();
        // ---- stage the halo tile, channels [c0, c0+CK): all loads in flight, masked when written
        {
            float4 r[ITERS];
            unsigned m[ITERS];
#pragma unroll
            for (int u = 0; u < ITERS; ++u) {
                const int e = tid + u * NTHR;
                const int pix = e / Q4, c4 = e - pix * Q4;
                const int hy = pix / TWH, hx = pix - hy * TWH;
                const int gy = y0 - PAD + hy, gx = x0 - PAD + hx;
                const bool ok = e < TOTAL && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
#ifdef STREAM_ABL_NO_STAGE
                r[u] = make_float4(1.f, 1.f, 1.f, 1.f);
#else
                r[u] = view_load4_raw(a.in, n, gy, gx, c0 + c4 * 4, ok && c0 + c4 * 4 < a.Cin);
#endif
                m[u] = valid4(c0 + c4 * 4, a.Cin, ok);
            }
#pragma unroll
            for (int u = 0; u < ITERS; ++u) {
                const int e = tid + u * NTHR;
                if (e < TOTAL) {
                    const int pix = e / Q4, c4 = e - pix * Q4;
                    const float4 v = mask4(r[u], m[u]);
                    float* d = tile + (size_t)pix * P + c4 * 4;
                    if (G == 4) {
                        *reinterpret_cast<float4*>(d) = v;
                    } else {
                        reinterpret_cast<float2*>(d)[0] = make_float2(v.x, v.y);
                        reinterpret_cast<float2*>(d)[1] = make_float2(v.z, v.w);
                    }
                }
            }
        }
        __syncthreads();
        STAMP();

        // ---- K loop over group-steps gs = tap*NGRP + g, software-pipelined: pixel fragments one group ahead,
        //      filter fragments WPD groups ahead (L2 latency)
        const int row_lane = c0 + E * lq;
        auto load_w = [&](int gs, wvec_t (&dst)[G]) {
            const int tap = gs / NGRP, g = gs - tap * NGRP;
#pragma unroll
            for (int s = 0; s < G; ++s) {
                int row = tap * a.Cin + row_lane + g * G + s;
#ifdef STREAM_ABL_NO_W
                row = row_lane + s;
#endif
                if (tap == KK - 1) row = min(row, last_row);
                dst[s] = *reinterpret_cast<const wvec_t*>(wlane + (size_t)row * sp.cw);
            }
        };
        auto load_a = [&](int gs, float (&dst)[MT][G]) {
            const int tap = gs / NGRP, g = gs - tap * NGRP;
            const int ky = tap / KS, kx = tap - ky * KS;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const float* src = rd + ((i + ky) * TWH + kx) * P + g * G;
                if (G == 4) {
                    const float4 v = *reinterpret_cast<const float4*>(src);
                    dst[i][0] = v.x; dst[i][1] = v.y; dst[i][2 % G] = v.z; dst[i][3 % G] = v.w;
                } else {
                    const float2 v = *reinterpret_cast<const float2*>(src);
                    dst[i][0] = v.x; dst[i][1] = v.y;
                }
            }
        };
        wvec_t wv[WPD + 1][G];
        float av[2][MT][G];
#pragma unroll
        for (int d = 0; d < WPD; ++d)
            if (d < NGS) load_w(d, wv[d]);
        load_a(0, av[0]);
#pragma unroll
        for (int gs = 0; gs < NGS; ++gs) {
            if (gs + WPD < NGS) load_w(gs + WPD, wv[(gs + WPD) % (WPD + 1)]);
            if (gs + 1 < NGS) load_a(gs + 1, av[(gs + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#ifdef STREAM_SETPRIO
            __builtin_amdgcn_s_setprio(STREAM_SETPRIO);
#endif
#pragma unroll
            for (int s = 0; s < G; ++s)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wget<NT>(wv[gs % (WPD + 1)][s], j), av[gs & 1][i][s],
                                                                          acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        STAMP();
    }

#ifdef STREAM_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
#ifdef STREAM_ABL_NO_EPI
    if (acc[0][0][0] != 12345.678f) return;
#endif
    // ---- epilogue: lane (pixel column l15, k-slot lq) owns couts n0 + 4*NT*lq + [0, 4*NT) of rows wave*4 + i
    const int gx = x0 + l15;
    const int cb = n0 + 4 * NT * lq;
    float4 bias_v[NT];
    size_t q_out[NT], q_add[NT], q_mask[NT];
    bool cok[NT];
#pragma unroll
    for (int v = 0; v < NT; ++v) {
        const int co = cb + 4 * v;
        cok[v] = co < a.Cout;
        const int cs = cok[v] ? co : 0;
        q_out[v] = view_chan_off(a.out, cs);
        q_add[v] = a.add.p ? view_chan_off(a.add, cs) : 0;
        q_mask[v] = a.mask.p ? view_chan_off(a.mask, cs) : 0;
        bias_v[v] = (a.bias && cok[v]) ? *reinterpret_cast<const float4*>(a.bias + cs) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int gy = y0 + wave * MT + i;
        if (gy < a.H && gx < a.W) {
            const size_t pb_out = view_pix_base(a.out, n, gy, gx);
            const size_t pb_add = a.add.p ? view_pix_base(a.add, n, gy, gx) : 0;
            const size_t pb_mask = a.mask.p ? view_pix_base(a.mask, n, gy, gx) : 0;
#pragma unroll
            for (int v = 0; v < NT; ++v) {
                if (cok[v]) {
                    // value index 4v+c within the lane's 4*NT couts  <->  accumulator (tile j, reg r): 4v+c = NT*r + j
                    float o[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) o[c] = acc[i][(4 * v + c) % NT][(4 * v + c) / NT];
                    float4 r4 = make_float4(o[0] + bias_v[v].x, o[1] + bias_v[v].y, o[2] + bias_v[v].z, o[3] + bias_v[v].w);
                    if (a.add.p) {
                        const float4 ad = *reinterpret_cast<const float4*>(a.add.p + pb_add + q_add[v]);
                        r4.x += ad.x; r4.y += ad.y; r4.z += ad.z; r4.w += ad.w;
                    }
                    if (a.relu) { r4.x = fmaxf(r4.x, 0.f); r4.y = fmaxf(r4.y, 0.f); r4.z = fmaxf(r4.z, 0.f); r4.w = fmaxf(r4.w, 0.f); }
                    if (a.mask.p) {
                        const float4 mk = *reinterpret_cast<const float4*>(a.mask.p + pb_mask + q_mask[v]);
                        r4.x = mk.x > 0.f ? r4.x : 0.f; r4.y = mk.y > 0.f ? r4.y : 0.f;
                        r4.z = mk.z > 0.f ? r4.z : 0.f; r4.w = mk.w > 0.f ? r4.w : 0.f;
                    }
                    float4* dst = reinterpret_cast<float4*>(a.out.p + pb_out + q_out[v]);
                    if (a.accumulate) {
                        const float4 old = *dst;
                        r4.x += old.x; r4.y += old.y; r4.z += old.z; r4.w += old.w;
                    }
                    *dst = r4;
                }
            }
        }
    }
    STAMP();
}

// ---------------------------------------------------------------------------------------------------------------------
// Producer / consumer form: ONE persistent 8-wave workgroup per CU.  Waves 0-3 (one per SIMD) issue nothing but the K
// loop -- filter fragments from L2, pixel fragments from LDS, MFMAs -- and hand their accumulators over through LDS
// (24 ds_write_b128 per item); waves 4-7 stage the NEXT chunk's halo tile into the other LDS buffer and run the epilogue
// of the PREVIOUS item (bias / add / ReLU / mask / store) out of that hand-over area while the matrix pipe keeps going.
// Why: with two independent 4-wave workgroups per CU the staging and epilogue phases of one workgroup are starved by the
// other's MFMAs (one VALU issue per MFMA slot, tools/ubench/mfma_valu.hip) and stretch 3-6x, and 18 % of the time neither
// workgroup is in its K loop (profiles/stream_trace_r02.txt).  Here the overlap is by construction: the helpers' work is
// free for the matrix pipe as long as it fits into one chunk's K loop.
// LDS: [tile A | spare | tile B]; the hand-over area of an item overlays the tile buffer its last chunk was read from
// plus the spare (the other buffer already holds the next step's tile), so the helpers must drain it before they stage
// the step after next into it -- a 4-wave counter barrier in LDS, the MFMA waves are not involved.
template <int KS, int E, int NT, int MT>
struct WsGeom {
    static constexpr int G = (E % 4 == 0) ? 4 : 2, CK = 4 * E, P = CK + G;
    static constexpr int TW = 16, TH = 4 * MT, TWH = TW + KS - 1, THH = TH + KS - 1, HPIX = TWH * THH;
    static constexpr int TILE = HPIX * P;                         // floats
    static constexpr int CO = 16 * NT, NPIX = TW * TH, DUMP = NPIX * CO;
    static constexpr int TOT = (2 * TILE > TILE + DUMP) ? 2 * TILE : TILE + DUMP;
    static constexpr size_t LDS_BYTES = (size_t)(TOT + 4) * 4;    // + the helpers' counter
    static constexpr bool fits = LDS_BYTES <= 160 * 1024 && TILE % 4 == 0;
};

typedef const char __attribute__((address_space(1)))* gcptr_t;
typedef char __attribute__((address_space(1)))* gptr_t;
typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef const f32x4 __attribute__((address_space(1)))* gcf4_t;     // (builtin vector type: loads / stores through an
typedef f32x4 __attribute__((address_space(1)))* gf4_t;            //  address-space pointer need no class copy constructor)
__device__ __forceinline__ float4 gload4(gcptr_t p) {
    const f32x4 v = *(gcf4_t)p;
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void gstore4(gptr_t p, const float4& v) { *(gf4_t)p = (f32x4){v.x, v.y, v.z, v.w}; }

// pixel strides (floats) of a view: pix_base(n, y, x) = n * nstride + y * sy + x * sx
__device__ __forceinline__ void view_strides(const TView& v, size_t& sy, size_t& sx) {
    const int r = v.d2s > 1 ? v.d2s : 1;
    sx = (size_t)r * v.ld;
    sy = (size_t)r * (size_t)(v.W * r) * v.ld;
}

// NL: views whose depth_to_space groups are narrower than a chunk / an n-block (per-quad channel offsets through the view)
template <int KS, int E, int NT, int MT, bool NL>
__global__ void __launch_bounds__(512, 1) conv_stream_ws_kernel(const StreamParams sp) {
    typedef WsGeom<KS, E, NT, MT> GM;
    const ConvParams& a = sp.c;
    constexpr int G = GM::G, NGRP = E / G, CK = GM::CK, P = GM::P;
    constexpr int TW = 16, TH = GM::TH, PAD = KS / 2, TWH = GM::TWH, THH = GM::THH, HPIX = GM::HPIX;
    constexpr int KK = KS * KS, NGS = KK * NGRP, Q4 = CK / 4;
    constexpr int HT = 256;                                       // helper threads
    constexpr int CO = GM::CO, NQ = 4 * NT;
    // filter prefetch distance in group-steps; the ring of WPD + 1 fragment sets keeps turning across steps, so its size must
    // divide the group-steps of a step (27 for 3x3; 75 or 50 for 5x5 with 24- / 32-channel chunks)
    constexpr int WPD = (NGS % 3 == 0) ? 2 : 1;
    // (an odd count that is no multiple of 3 -- 25 for 5x5 with 16-channel chunks -- is padded with one idle group-step)
    constexpr int NGSP = (NGS + WPD) / (WPD + 1) * (WPD + 1);
    static_assert(NGSP % (WPD + 1) == 0 && NGS > WPD && NGSP - NGS <= 1, "the filter ring runs across steps");
    typedef typename WVec<NT>::T wvec_t;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const bufA = lds;
    float* const bufB = lds + (GM::TOT - GM::TILE);
    unsigned* const ctr = reinterpret_cast<unsigned*>(lds + GM::TOT);        // [0]: helpers' barrier, [1]: accumulators handed over

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave8 = tid >> 6;
#ifdef WS_HELPERS_LAST
    const bool helper = wave8 >= 4;
#else
    const bool helper = wave8 < 4;                                // (the OLDER waves of each SIMD: see below)
#endif
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, SX = gridDim.x >> 3;
    const int nchunks = a.Cin / CK;                               // (the launcher guarantees Cin % CK == 0)
    const int tiles_here = min(sp.per_xcd, sp.ntiles - xcd * sp.per_xcd);
    const int nvalid = max(tiles_here, 0) * sp.nblk;              // this XCD's items: (tile, n-block), n-blocks innermost
    if (tid < 2) ctr[tid] = 0u;

#ifdef STREAM_TRACE
    // diagnostics build: per-workgroup sums of phase times (100 MHz ticks; word 0 in shader cycles), tools/ws_trace.py
    unsigned long long tr_t = 0, tr_t0 = 0, tr_c = 0;
    unsigned long long* const trw = sp.trace ? sp.trace + (size_t)blockIdx.x * 16 : nullptr;
    const bool tr_on = trw && lane == 0 && (wave8 == 0 || wave8 == 4);      // (one MFMA wave, one helper wave)
    if (trw && lane == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
        atomicOr(trw + 13, (unsigned long long)((hw >> 4) & 3u) << (2 * wave8));
        if (wave8 == 0) trw[6] = hw;
    }
    unsigned long long tr_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // (sums stay in registers until the end)
#define WS_T0() do { tr_t0 = wall_clock64(); } while (0)
#define WS_TIC() do { tr_t = wall_clock64(); } while (0)
#define WS_TOC(slot_) do { const unsigned long long n_ = wall_clock64(); tr_acc[slot_] += n_ - tr_t; tr_t = n_; } while (0)
#define WS_END(slot_) do { tr_acc[slot_] = wall_clock64() - tr_t0; if (tr_on) { for (int q_ = 0; q_ < 16; ++q_) if (q_ != 6 && q_ != 13 && tr_acc[q_]) trw[q_] = tr_acc[q_]; } } while (0)
#define WS_CYC0() do { tr_c = clock64(); } while (0)
#define WS_CYC1() do { tr_acc[0] += clock64() - tr_c; } while (0)
#else
#define WS_T0()
#define WS_TIC()
#define WS_TOC(slot_)
#define WS_END(slot_)
#define WS_CYC0()
#define WS_CYC1()
#endif
    struct Item { int n, y0, x0, n0; };
    auto decode = [&](int li) {
        const int tl = fast_div(li, sp.m_nblk);
        const int nb = li - tl * sp.nblk;
        const int t = xcd * sp.per_xcd + tl;
        const int q = fast_div(t, a.m_txy[0]);
        const int bx = t - q * a.tiles_x;
        const int n = fast_div(q, a.m_txy[1]);
        const int by = q - n * a.tiles_y;
        Item it;
        it.n = n; it.y0 = by * TH; it.x0 = bx * TW; it.n0 = nb * 16 * NT;
        return it;
    };

    if (helper) {
        // A helper wave gets an instruction issued only every ~80 cycles while the wave it shares its SIMD with streams MFMAs
        // (measured, tools/ws_trace.py) -- of ANY kind, not only VALU.  So per element there is one memory instruction and
        // nothing else: the offsets of a thread's elements relative to the tile origin are computed once per kernel, the
        // origin is a wave-uniform buffer descriptor (scalar unit), and whatever lies outside the image or the tensor gets
        // an out-of-range offset instead of a branch or a select: the buffer unit returns zeros for such loads -- exactly the
        // convolution's zero padding -- and drops such stores.
        const int htid = tid & 255;
        constexpr int OOB = (int)0xffffff00u;
        constexpr int RSRC3 = 0x00020000;                         // gfx9 raw buffer, 32-bit data
        // ---- staging: thread = (channel quad c4, pixel p0 + PPASS * u)
        constexpr int PPASS = HT / Q4, SIT = (HPIX + PPASS - 1) / PPASS;
        const int c4 = htid % Q4, p0 = htid / Q4;
        const bool st_active = p0 < PPASS;
        size_t isy, isx;
        view_strides(a.in, isy, isx);
        // channel offset of the thread's quad inside a chunk: linear views -> 4*c4 on top of the chunk's own offset (part of the
        // origin); through a view whose depth_to_space groups are narrower than a chunk the per-quad offsets go through the
        // view and depend on the chunk, which then is part of the signature below (recomputed per chunk)
        const bool in_lin = !NL || a.in.d2s <= 1 || (CK <= a.in.cp && a.in.cp % CK == 0);
        size_t in_c4 = in_lin ? (size_t)c4 * 4 : view_chan_off(a.in, min(c4 * 4, a.Cin - 4));
        int soff[SIT], hyx[SIT];
        auto rel_of = [&](int hy, int hx) { return (int)((hy * isy + hx * isx + in_c4) * 4); };
#pragma unroll
        for (int u = 0; u < SIT; ++u) {
            const int hp = p0 + PPASS * u;
            const int hy = hp / TWH, hx = hp - hy * TWH;
            const bool live = st_active && hp < HPIX;
            hyx[u] = live ? ((hy << 8) | hx) : 0x7f7f;            // (0x7f: never inside the image window)
            soff[u] = live ? rel_of(hy, hx) : OOB;
        }
        int st_sig = (THH << 8) | TWH;                            // window signature of soff[]: (ylo, yhi, xlo, xhi) packed
        st_sig |= 0 << 24;
        const int st_dst = (p0 * P + c4 * 4);                     // floats; element u goes PPASS * P * u further
        auto stage = [&](const Item& it, int c0, float* tile) __attribute__((always_inline)) {
            // the part of the halo tile that lies inside the image: rows [ylo, yhi), columns [xlo, xhi)
            const int ylo = max(0, PAD - it.y0), yhi = min(THH, a.H + PAD - it.y0);
            const int xlo = max(0, PAD - it.x0), xhi = min(TWH, a.W + PAD - it.x0);
            const int sig = ((in_lin ? 0 : c0 / CK) << 26) | (ylo << 24) | (xlo << 16) | (yhi << 8) | xhi;
            if (sig != st_sig) {                                  // (wave-uniform; n-blocks and chunks of one tile share it)
                st_sig = sig;
                if (!in_lin) in_c4 = view_chan_off(a.in, min(c0 + c4 * 4, a.Cin - 4));
#pragma unroll
                for (int u = 0; u < SIT; ++u) {
                    const int hy = hyx[u] >> 8, hx = hyx[u] & 0xff;
                    soff[u] = (hy >= ylo && hy < yhi && hx >= xlo && hx < xhi) ? rel_of(hy, hx) : OOB;
                }
            }
            // wave-uniform origin of the halo tile (outside the tensor for border tiles; never dereferenced there)
            const long org = (long)((size_t)it.n * a.in.nstride) + (long)(it.y0 - PAD) * (long)isy + (long)(it.x0 - PAD) * (long)isx +
                             (in_lin ? (long)view_chan_off(a.in, c0) : 0l);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char*>(reinterpret_cast<const char*>(a.in.p)) + org * 4, 0, 0x7fffff00, RSRC3);
            float* dst = tile + st_dst;
            i32x4_t r[SIT];
#pragma unroll
            for (int u = 0; u < SIT; ++u) r[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, soff[u], 0, 0);
#ifdef STREAM_TRACE
            WS_TOC(14);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            WS_TOC(15);
#endif
#pragma unroll
            for (int u = 0; u < SIT; ++u) {
                if (u + 1 < SIT ? (PPASS * Q4 == HT || st_active) : (hyx[u] >> 8) < THH) {
                    float* d = dst + u * (PPASS * P);
                    if (G == 4) {
                        *reinterpret_cast<i32x4_t*>(d) = r[u];
                    } else {
                        reinterpret_cast<int2*>(d)[0] = make_int2(r[u][0], r[u][1]);
                        reinterpret_cast<int2*>(d)[1] = make_int2(r[u][2], r[u][3]);
                    }
                }
            }
        };
        // ---- epilogue out of the hand-over area [pixel][CO] (linear): thread owns elements e = htid + 256 * u, element =
        //      (pixel e / NQ, channel quad e % NQ) -> a wave stores whole 16*NQ-byte pixel segments
        constexpr int ND = MT * NT;                               // elements per thread
        size_t osy, osx;
        view_strides(a.out, osy, osx);                            // (add / mask views have the same strides: launcher)
        // (the same for the output's channel quads inside an n-block; otherwise they go through the view per n-block)
        const bool out_lin = !NL || a.out.d2s <= 1 || (CO <= a.out.cp && a.out.cp % CO == 0);
        int dvo[ND];
#pragma unroll
        for (int u = 0; u < ND; ++u) {
            const int e = htid + HT * u;
            const int pix = e / NQ, quad = e - pix * NQ;
            dvo[u] = (int)(((pix >> 4) * osy + (pix & 15) * osx + (out_lin ? (size_t)quad * 4 : view_chan_off(a.out, min(4 * quad, a.Cout - 4)))) * 4);
        }
        int dr_sig = (NQ << 16) | (TH << 8) | TW;                 // signature of dvo[]: (n-block,) quads, rows, columns that exist
        auto drain = [&](const Item& it, const float* dump) __attribute__((always_inline)) {
            const int nq = min(NQ, (a.Cout - it.n0) >> 2);        // channel quads of this n-block that exist
            const int ymax = min(TH, a.H - it.y0), xmax = min(TW, a.W - it.x0);
            const int sig = (out_lin ? 0 : it.n0 << 24) | (nq << 16) | (ymax << 8) | xmax;
            if (sig != dr_sig) {                                  // (wave-uniform: ragged image edges, last n-block of a ragged Cout)
                dr_sig = sig;
#pragma unroll
                for (int u = 0; u < ND; ++u) {
                    const int e = htid + HT * u;
                    const int pix = e / NQ, quad = e - pix * NQ;
                    const size_t ch = out_lin ? (size_t)quad * 4 : view_chan_off(a.out, min(it.n0 + 4 * quad, a.Cout - 4));
                    const int o = (int)(((pix >> 4) * osy + (pix & 15) * osx + ch) * 4);
                    dvo[u] = ((pix >> 4) < ymax && (pix & 15) < xmax && quad < nq) ? o : OOB;
                }
            }
            const size_t ooff = (size_t)it.n * a.out.nstride + it.y0 * osy + it.x0 * osx + (out_lin ? view_chan_off(a.out, it.n0) : 0);
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.out.p) + ooff * 4, 0, 0x7fffff00, RSRC3);
            const size_t aoff = a.add.p ? (size_t)it.n * a.add.nstride + it.y0 * osy + it.x0 * osx + (out_lin ? view_chan_off(a.add, it.n0) : 0) : 0;
            const size_t moff = a.mask.p ? (size_t)it.n * a.mask.nstride + it.y0 * osy + it.x0 * osx + (out_lin ? view_chan_off(a.mask, it.n0) : 0) : 0;
            const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.add.p) + aoff * 4, 0, 0x7fffff00, RSRC3);
            const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.mask.p) + moff * 4, 0, 0x7fffff00, RSRC3);
            const float* src = dump + htid * 4;
            // every combination of the epilogue switches gets its own straight-line code, in groups of GQ elements: all loads
            // of a group are issued before the first value is used
            auto body = [&](auto mode_tag) __attribute__((always_inline)) {
                constexpr int M = decltype(mode_tag)::value;
                constexpr bool ADD = (M & 1) != 0, RELU = (M & 2) != 0, MASK = (M & 4) != 0, ACC = (M & 8) != 0;
                constexpr int NSTREAM = 1 + (ADD ? 1 : 0) + (MASK ? 1 : 0) + (ACC ? 1 : 0);
                constexpr int GQ = (NSTREAM <= 1 && ND % 6 == 0) ? 6 : 4;
                static_assert(ND % GQ == 0, "elements per thread");
#pragma unroll
                for (int g0 = 0; g0 < ND; g0 += GQ) {
                    f32x4 v[GQ];
                    i32x4_t ad[ADD ? GQ : 1], mk[MASK ? GQ : 1], old[ACC ? GQ : 1];
#pragma unroll
                    for (int i = 0; i < GQ; ++i) v[i] = *reinterpret_cast<const f32x4*>(src + (g0 + i) * (HT * 4));
                    if (ADD) {
#pragma unroll
                        for (int i = 0; i < GQ; ++i) ad[i] = __builtin_amdgcn_raw_buffer_load_b128(ra, dvo[g0 + i], 0, 0);
                    }
                    if (MASK) {
#pragma unroll
                        for (int i = 0; i < GQ; ++i) mk[i] = __builtin_amdgcn_raw_buffer_load_b128(rm, dvo[g0 + i], 0, 0);
                    }
                    if (ACC) {
#pragma unroll
                        for (int i = 0; i < GQ; ++i) old[i] = __builtin_amdgcn_raw_buffer_load_b128(ro, dvo[g0 + i], 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < GQ; ++i) {
                        f32x4 r4 = v[i];
                        if (ADD) r4 += __builtin_bit_cast(f32x4, ad[i]);
                        if (RELU) { r4[0] = fmaxf(r4[0], 0.f); r4[1] = fmaxf(r4[1], 0.f); r4[2] = fmaxf(r4[2], 0.f); r4[3] = fmaxf(r4[3], 0.f); }
                        if (MASK) {
                            const f32x4 m = __builtin_bit_cast(f32x4, mk[i]);
                            r4[0] = m[0] > 0.f ? r4[0] : 0.f; r4[1] = m[1] > 0.f ? r4[1] : 0.f;
                            r4[2] = m[2] > 0.f ? r4[2] : 0.f; r4[3] = m[3] > 0.f ? r4[3] : 0.f;
                        }
                        if (ACC) r4 += __builtin_bit_cast(f32x4, old[i]);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, r4), ro, dvo[g0 + i], 0, 0);
                    }
                }
            };
            const int mode = (a.add.p ? 1 : 0) | (a.relu ? 2 : 0) | (a.mask.p ? 4 : 0) | (a.accumulate ? 8 : 0);
            if (mode == 0) body(std::integral_constant<int, 0>{});
            else if (mode == 1) body(std::integral_constant<int, 1>{});
            else if (mode == 2) body(std::integral_constant<int, 2>{});
            else if (mode == 3) body(std::integral_constant<int, 3>{});
            else if (mode == 4) body(std::integral_constant<int, 4>{});
            else if (mode == 5) body(std::integral_constant<int, 5>{});
            else if (mode == 6) body(std::integral_constant<int, 6>{});
            else if (mode == 7) body(std::integral_constant<int, 7>{});
            else if (mode == 8) body(std::integral_constant<int, 8>{});
            else if (mode == 9) body(std::integral_constant<int, 9>{});
            else if (mode == 10) body(std::integral_constant<int, 10>{});
            else if (mode == 11) body(std::integral_constant<int, 11>{});
            else if (mode == 12) body(std::integral_constant<int, 12>{});
            else if (mode == 13) body(std::integral_constant<int, 13>{});
            else if (mode == 14) body(std::integral_constant<int, 14>{});
            else body(std::integral_constant<int, 15>{});
        };

        unsigned epoch = 0, handed = 0;
        auto helper_sync = [&]() {                                // the four helper waves only; the MFMA waves are in their K loop
            ++epoch;
            if (lane == 0) {
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 4u * epoch)
                    __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_wave_barrier();
        };
        auto wait_handed = [&]() {                                // all four MFMA waves have written their accumulators
            ++handed;
            if (lane == 0) {
                while (__hip_atomic_load(ctr + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 4u * handed)
                    __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_wave_barrier();
        };
        // one loop over "steps" (item, chunk); iteration k runs beside the MFMA waves' K loop of step k and prepares step k+1
        // (iteration -1 = the prologue: stage step 0)
        bool pending = false;
        Item pend_it = {0, 0, 0, 0}, cur = {0, 0, 0, 0};
        const float* pend_dump = nullptr;
        int li = slot, c = -1;                                    // step k = (li, c)
        WS_T0();
        for (int k = -1;; ++k) {
            if (pending) {
                WS_TIC();
                wait_handed();
                drain(pend_it, pend_dump);
                WS_TOC(8);
                helper_sync();
                WS_TOC(9);
                pending = false;
            }
            WS_TIC();
            int nli = li, nc = c + 1;
            if (k >= 0 && nc == nchunks) { nli = li + SX; nc = 0; }
            const bool has_next = nli < nvalid;
            Item nxt = cur;
            if (has_next) {
                if (nc == 0) nxt = decode(nli);
                stage(nxt, nc * CK, ((k + 1) & 1) ? bufB : bufA);
            }
            WS_TOC(10);
            __syncthreads();                                      // S0 (k = -1) / X: tile k consumed, tile k+1 staged
            if (k >= 0 && c + 1 == nchunks) {
                pending = true;
                pend_it = cur;
                pend_dump = (k & 1) ? lds + (GM::TOT - GM::DUMP) : lds;
            }
            WS_TOC(11);
            if (!has_next) break;
            cur = nxt; li = nli; c = nc;
        }
        if (pending) { wait_handed(); drain(pend_it, pend_dump); }
        WS_END(12);
        return;
    }

    // ---------------- MFMA waves
    const int wave = wave8 & 3;
    const int l15 = lane & 15, lq = lane >> 4;
    const int rd_off = ((wave * MT) * TWH + l15) * P + E * lq;
    const unsigned row_bytes = (unsigned)sp.cw * 4u;
    const unsigned tap_bytes = (unsigned)a.Cin * row_bytes;
    const unsigned tap_step = tap_bytes - (unsigned)(E - 1) * row_bytes;
    typedef const wvec_t __attribute__((address_space(1)))* gvec_t;
    // filter: lane (row l15, k-slot lq) reads floats [co, co+NT) of row tap*Cin + c*CK + E*lq + e; the lane part of the address
    // is one 32-bit offset, the (tap, e) part a wave-uniform base pointer that walks the rows in load order (made opaque at
    // every load so that it stays ONE SGPR pair advanced by s_add instead of 54 precomputed 64-bit vector addresses)
    auto lane_off = [&](int li_, int c_) {
        const int tl = fast_div(li_, sp.m_nblk);
        const int n0 = (li_ - tl * sp.nblk) * 16 * NT;
        const int co_lane = min(n0 + NT * l15, sp.cw - NT);
        return (unsigned)co_lane * 4u + (unsigned)(c_ * CK + E * lq) * row_bytes;
    };
    // the accumulators start at the bias: lane (pixel column l15, k-slot lq) owns couts n0 + 4*NT*lq + [0, 4*NT)
    auto load_bias = [&](int li_, float4 (&bv)[NT]) {
        const int tl = fast_div(li_, sp.m_nblk);
        const int cb = (li_ - tl * sp.nblk) * 16 * NT + 4 * NT * lq;
#pragma unroll
        for (int v = 0; v < NT; ++v)
            bv[v] = (a.bias && cb + 4 * v < a.Cout) ? *reinterpret_cast<const float4*>(a.bias + cb + 4 * v) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    f32x4 acc[MT][NT];
    auto init_acc = [&](const float4 (&bv)[NT]) {
#pragma unroll
        for (int v = 0; v < NT; ++v) {
            const float o[4] = {bv[v].x, bv[v].y, bv[v].z, bv[v].w};
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i][(4 * v + cc) % NT][(4 * v + cc) / NT] = o[cc];
        }
    };
    gcptr_t wp = (gcptr_t)(reinterpret_cast<const char*>(a.w));
    unsigned voff = 0;
    auto load_w = [&](int e, wvec_t (&dst)[G]) {                  // e = first k-slot row of the group within its tap
#pragma unroll
        for (int s = 0; s < G; ++s) {
            asm volatile("" : "+s"(wp));
            dst[s] = *(gvec_t)(wp + (size_t)voff);
            wp += (e + s == E - 1) ? tap_step : row_bytes;
        }
    };
    wvec_t wv[WPD + 1][G];
    float4 bnext[NT];
    if (slot < nvalid) {
        load_bias(slot, bnext);
        voff = lane_off(slot, 0);
#pragma unroll
        for (int d = 0; d < WPD; ++d) load_w((d % NGRP) * G, wv[d]);
    } else {
#pragma unroll
        for (int v = 0; v < NT; ++v) bnext[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    init_acc(bnext);

    __syncthreads();                                              // S0
    int k = 0;
    WS_T0();
    for (int li = slot; li < nvalid; li += SX) {
        for (int c = 0; c < nchunks; ++c) {
            const float* rd = ((k & 1) ? bufB : bufA) + rd_off;
            const bool last = c + 1 == nchunks;
            // the step after this one (its first filter fragments are requested at the tail of this K loop)
            const int nli = last ? li + SX : li, nc = last ? 0 : c + 1;
            const unsigned voff_next = nli < nvalid ? lane_off(nli, nc) : voff;
            if (last && nli < nvalid) load_bias(nli, bnext);
            auto load_a = [&](int gs, float (&dst)[MT][G]) {
                const int tap = gs / NGRP, g = gs - tap * NGRP;
                const int ky = tap / KS, kx = tap - ky * KS;
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const float* src = rd + ((i + ky) * TWH + kx) * P + g * G;
                    if (G == 4) {
                        const float4 v = *reinterpret_cast<const float4*>(src);
                        dst[i][0] = v.x; dst[i][1] = v.y; dst[i][2 % G] = v.z; dst[i][3 % G] = v.w;
                    } else {
                        const float2 v = *reinterpret_cast<const float2*>(src);
                        dst[i][0] = v.x; dst[i][1] = v.y;
                    }
                }
            };
            float av[2][MT][G];
            WS_TIC();
            WS_CYC0();
            load_a(0, av[0]);
#pragma unroll
            for (int gs = 0; gs < NGSP; ++gs) {
                if (gs + WPD == NGSP) {                           // from here on: the next step's rows
                    wp = (gcptr_t)(reinterpret_cast<const char*>(a.w));
                    voff = voff_next;
                }
                if ((gs + WPD) % NGSP < NGS) load_w((((gs + WPD) % NGSP) % NGRP) * G, wv[(gs + WPD) % (WPD + 1)]);
                if (gs + 1 < NGS) load_a(gs + 1, av[(gs + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                if (gs < NGS) {
#pragma unroll
                    for (int s = 0; s < G; ++s)
#pragma unroll
                        for (int i = 0; i < MT; ++i)
#pragma unroll
                            for (int j = 0; j < NT; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wget<NT>(wv[gs % (WPD + 1)][s], j), av[gs & 1][i][s],
                                                                                  acc[i][j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            WS_CYC1();
            WS_TOC(1);
            __syncthreads();                                      // X
            WS_TOC(2);
#ifdef STREAM_TRACE
            tr_acc[5] += 1;
#endif
            if (last) {
                // hand the accumulators over: lane (pixel column l15, k-slot lq) owns couts 4*NT*lq + [0, 4*NT) of rows wave*MT + i
                float* dump = ((k & 1) ? lds + (GM::TOT - GM::DUMP) : lds) + (wave * MT * 16 + l15) * CO + 4 * NT * lq;
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int v = 0; v < NT; ++v) {
                        float4 o;
                        o.x = acc[i][(4 * v + 0) % NT][(4 * v + 0) / NT];
                        o.y = acc[i][(4 * v + 1) % NT][(4 * v + 1) / NT];
                        o.z = acc[i][(4 * v + 2) % NT][(4 * v + 2) / NT];
                        o.w = acc[i][(4 * v + 3) % NT][(4 * v + 3) / NT];
                        *reinterpret_cast<float4*>(dump + i * 16 * CO + 4 * v) = o;
                    }
                init_acc(bnext);
                if (lane == 0) __hip_atomic_fetch_add(ctr + 1, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                WS_TOC(3);
            }
            ++k;
        }
    }
    WS_END(4);
}

// Output-channel counts that no NT divides (40 = RB5 of the headline backbone) used to fall back to NT = 1: 16 couts per
// block, one dword of filter per lane and k-step, 65 TFLOP/s.  They now run with the widest NT that pads Cout no further
// than NT = 1 would (40 -> 48 with NT = 3) on a zero-padded copy of the filter ([rows][cw], cw = whole blocks), so
// a lane's NT consecutive couts never straddle the end of a row; the padded columns are computed and never stored.
__global__ void pad_filter_kernel(const float* __restrict__ w, float* __restrict__ wp, int rows, int Cout, int cw) {
    const int total = rows * cw;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int r = e / cw, c = e - r * cw;
        wp[e] = c < Cout ? w[(size_t)r * Cout + c] : 0.f;
    }
}
struct PadScratch { hipStream_t stream; float* buf; size_t floats; };
float* pad_scratch(hipStream_t s, size_t floats) {       // grow-only, one buffer per stream (launches on a stream are ordered)
    static std::mutex mu;
    static std::vector<PadScratch> all;
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
    PadScratch e{s, nullptr, std::max<size_t>(floats, 1 << 16)};
    HIP_CHECK(hipMalloc((void**)&e.buf, e.floats * sizeof(float)));
    all.push_back(e);
    return e.buf;
}

template <int KS, int E, int NT, int MT>
void launch_stream(hipStream_t s, StreamParams& sp, int N) {
    constexpr int G = (E % 4 == 0) ? 4 : 2;
    constexpr int P = 4 * E + G;
    constexpr int HPIX = (16 + KS - 1) * (4 * MT + KS - 1);
    constexpr size_t lds = (size_t)HPIX * P * sizeof(float);
    ConvParams& p = sp.c;
    p.tiles_x = cdiv(p.W, 16);
    p.tiles_y = cdiv(p.H, 4 * MT);
    p.m_txy[0] = div_magic(p.tiles_x);
    p.m_txy[1] = div_magic(p.tiles_y);
    sp.ntiles = p.tiles_x * p.tiles_y * N;
    sp.nblk = cdiv(p.Cout, 16 * NT);
    sp.m_nblk = div_magic(sp.nblk);
    sp.cw = p.Cout;
    if (p.Cout % NT) {
        sp.cw = sp.nblk * 16 * NT;
        const int rows = KS * KS * p.Cin;
        float* wp = pad_scratch(s, (size_t)rows * sp.cw);
        ProfScope pp(s, "pad_filter", 0.0, 4.0 * rows * (p.Cout + sp.cw));
        DL4DS_LAUNCH(pad_filter_kernel, dim3(std::min(cdiv(rows * sp.cw, 256), 1024)), dim3(256), 0, s, p.w, wp, rows, p.Cout, sp.cw);
        HIP_CHECK(hipGetLastError());
        p.w = wp;
    }
    sp.per_xcd = cdiv(sp.ntiles, 8);
    auto kern = conv_stream_kernel<KS, E, NT, MT>;
    static std::once_flag once;
    std::call_once(once, [&]() {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds));
    });
    const unsigned grid = (unsigned)(8 * sp.per_xcd * sp.nblk);
    const double px = (double)N * p.H * p.W;
    ProfScope ps(s, "conv_stream<" + std::to_string(KS) + "," + std::to_string(E) + "," + std::to_string(NT) + "," +
                        std::to_string(MT) + ">",
                 2.0 * px * KS * KS * p.Cin * p.Cout,
                 4.0 * (px * (p.Cin + p.Cout * (1 + (p.add.p ? 1 : 0) + (p.mask.p ? 1 : 0) + (p.accumulate ? 1 : 0))) +
                        (double)KS * KS * p.Cin * p.Cout));
#ifdef STREAM_TRACE
    // -DSTREAM_TRACE build (tools/variant_build.sh): the first launches of the tall NT = 3 variant dump their timeline
    static unsigned long long* trace_buf = nullptr;
    static int trace_n = 0;
    sp.trace = nullptr;
    if (MT == 8 && NT == 3 && trace_n < 4 && grid <= 65536) {
        if (!trace_buf) HIP_CHECK(hipMalloc((void**)&trace_buf, (size_t)65536 * 8 * 8));
        HIP_CHECK(hipMemsetAsync(trace_buf, 0, (size_t)65536 * 8 * 8, s));
        sp.trace = trace_buf;
    }
#endif
    DL4DS_LAUNCH(kern, dim3(grid), dim3(kStreamThreads), lds, s, sp);
    HIP_CHECK(hipGetLastError());
#ifdef STREAM_TRACE
    if (sp.trace) {
        HIP_CHECK(hipStreamSynchronize(s));
        std::vector<unsigned long long> h((size_t)grid * 8);
        HIP_CHECK(hipMemcpy(h.data(), trace_buf, h.size() * 8, hipMemcpyDeviceToHost));
        char name[128];
        std::snprintf(name, sizeof name, "gpurun_out/stream_trace_%d.bin", trace_n++);
        if (FILE* f = std::fopen(name, "wb")) { std::fwrite(h.data(), 8, h.size(), f); std::fclose(f); }
    }
#endif
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

// producer / consumer form (conv_stream_ws_kernel): one persistent workgroup per CU
template <int KS, int E, int NT, int MT>
bool launch_stream_ws(hipStream_t s, StreamParams& sp, int N) {
    typedef WsGeom<KS, E, NT, MT> GM;
    if constexpr (!GM::fits) {
        return false;
    } else {
        ConvParams& p = sp.c;
        if (p.Cin % GM::CK) return false;
        // the helpers address a whole chunk / n-block with ONE origin plus per-thread offsets: channel offsets must be linear
        // inside a chunk (input) and an n-block (output), i.e. depth_to_space groups may not be straddled, and the add / mask
        // operands must be laid out like the output
        auto linear = [](const TView& v, int span) { return v.d2s <= 1 || (span <= v.cp && v.cp % span == 0); };
        auto same_layout = [](const TView& u, const TView& v) { return u.ld == v.ld && u.d2s == v.d2s && u.W == v.W && u.cp == v.cp; };
        const bool nl = !linear(p.in, GM::CK) || !linear(p.out, GM::CO);
        if (nl && MT != 4) return false;                                   // (only the 16x16-tile variants are built for them)
        if (!linear(p.in, GM::CK) && p.Cin / GM::CK > 31) return false;    // (the chunk index is part of the staging signature: 5 bits)
        if (!linear(p.out, GM::CO) && p.Cout >= 128) return false;         // (n0 must fit the 8 bits it gets in the drain signature)
        if (p.add.p && !same_layout(p.add, p.out)) return false;
        if (p.mask.p && !same_layout(p.mask, p.out)) return false;
        if ((size_t)(4 * MT + 2) * p.out.W * std::max(p.out.d2s, 1) * std::max(p.out.d2s, 1) * p.out.ld * 4 >= (1ull << 31)) return false;
        if ((size_t)(4 * MT + 2) * p.in.W * std::max(p.in.d2s, 1) * std::max(p.in.d2s, 1) * p.in.ld * 4 >= (1ull << 31)) return false;
        p.tiles_x = cdiv(p.W, 16);
        p.tiles_y = cdiv(p.H, 4 * MT);
        p.m_txy[0] = div_magic(p.tiles_x);
        p.m_txy[1] = div_magic(p.tiles_y);
        sp.ntiles = p.tiles_x * p.tiles_y * N;
        sp.nblk = cdiv(p.Cout, 16 * NT);
        sp.m_nblk = div_magic(sp.nblk);
        sp.per_xcd = cdiv(sp.ntiles, 8);
        const char* force = test_env("DL4DS_STREAM_FORCE_WS");       // (tests: "<workgroups per XCD>", small grids too)
        const int SX = force ? std::max(atoi(force), 1) : std::max(cu_count() / 8, 1);
        // at least one item per workgroup (two until round 3: "nothing to overlap" -- but the LDS-staged kernel these layers fell back to
        // is slower still: cfg5 568 -> 580 samples/s with 1, 581 with 0.5, 575 with 0.25; DL4DS_STREAM_MIN_ITEMS=<f> for A/B)
        static const double min_items = exp_env("DL4DS_STREAM_MIN_ITEMS") ? atof(exp_env("DL4DS_STREAM_MIN_ITEMS")) : 1.0;
        if (!force && (double)sp.per_xcd * sp.nblk < min_items * SX) return false;
        sp.cw = p.Cout;
        if (p.Cout % NT) {
            sp.cw = sp.nblk * 16 * NT;
            const int rows = KS * KS * p.Cin;
            float* wp = pad_scratch(s, (size_t)rows * sp.cw);
            ProfScope pp(s, "pad_filter", 0.0, 4.0 * rows * (p.Cout + sp.cw));
            DL4DS_LAUNCH(pad_filter_kernel, dim3(std::min(cdiv(rows * sp.cw, 256), 1024)), dim3(256), 0, s, p.w, wp, rows, p.Cout, sp.cw);
            HIP_CHECK(hipGetLastError());
            p.w = wp;
        }
        void (*kern)(const StreamParams) = conv_stream_ws_kernel<KS, E, NT, MT, false>;
        if constexpr (MT == 4) {
            if (nl) kern = conv_stream_ws_kernel<KS, E, NT, MT, true>;
        }
        static std::once_flag once;
        std::call_once(once, [&]() {
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_stream_ws_kernel<KS, E, NT, MT, false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)GM::LDS_BYTES));
            if constexpr (MT == 4)
                HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_stream_ws_kernel<KS, E, NT, MT, true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)GM::LDS_BYTES));
        });
        const double px = (double)N * p.H * p.W;
        ProfScope ps(s, "conv_stream_ws<" + std::to_string(KS) + "," + std::to_string(E) + "," + std::to_string(NT) + "," +
                            std::to_string(MT) + ">",
                     2.0 * px * KS * KS * p.Cin * p.Cout,
                     // (the fused epilogue's operands are part of the layer's algorithmic traffic: residual / ReLU-mask / old value)
                     4.0 * (px * (p.Cin + p.Cout * (1 + (p.add.p ? 1 : 0) + (p.mask.p ? 1 : 0) + (p.accumulate ? 1 : 0))) +
                            (double)KS * KS * p.Cin * p.Cout));
#ifdef STREAM_TRACE
        static unsigned long long* trace_buf = nullptr;
        static int trace_n = 0;
        sp.trace = nullptr;
        if (trace_n < 6) {
            if (!trace_buf) HIP_CHECK(hipMalloc((void**)&trace_buf, (size_t)4096 * 16 * 8));
            HIP_CHECK(hipMemsetAsync(trace_buf, 0, (size_t)4096 * 16 * 8, s));
            sp.trace = trace_buf;
        }
#endif
        DL4DS_LAUNCH(kern, dim3(8 * SX), dim3(512), GM::LDS_BYTES, s, sp);
        HIP_CHECK(hipGetLastError());
#ifdef STREAM_TRACE
        if (sp.trace) {
            HIP_CHECK(hipStreamSynchronize(s));
            std::vector<unsigned long long> h((size_t)8 * SX * 16);
            HIP_CHECK(hipMemcpy(h.data(), trace_buf, h.size() * 8, hipMemcpyDeviceToHost));
            char name[128];
            std::snprintf(name, sizeof name, "gpurun_out/ws_trace_%d.bin", trace_n++);
            if (FILE* f = std::fopen(name, "wb")) { std::fwrite(h.data(), 8, h.size(), f); std::fclose(f); }
        }
#endif
        return true;
    }
}

template <int KS, int E>
void dispatch_nt(hipStream_t s, StreamParams& sp, int N, int NT) {
    if constexpr (KS == 3 && E >= 6) {
        static const bool use_ws = exp_env("DL4DS_STREAM_NO_WS") == nullptr;
        if (use_ws && (NT >= 2 || (NT == 1 && E == 8 && !exp_env("DL4DS_STREAM_NO_WS_NT1")))) {
            const float* w0 = sp.c.w;
            bool done = false;
            if (NT == 1) { if constexpr (E == 8) done = launch_stream_ws<3, 8, 1, 4>(s, sp, N); }
            else if (NT == 2) done = launch_stream_ws<3, E, 2, 4>(s, sp, N);
            else if (NT == 3) done = launch_stream_ws<3, E, 3, 4>(s, sp, N);
            else done = launch_stream_ws<3, E, 4, 4>(s, sp, N);
            if (done) return;
            sp.c.w = w0;
        }
    }
    if constexpr (KS == 1 && E > 4) {      // (16-channel chunks: a single group-step per tile, the filter ring needs two)
        // 1x1 layers on the producer / consumer kernel too (round 3): the helper waves keep the next tile in flight while the MFMA
        // waves work -- 48 -> 48 at 64 x 128^2: 0.137 -> 0.110 ms (2.9 -> 3.7 TB/s).  DL4DS_STREAM_NO_WS1=1 for A/B.
        static const bool ws1 = exp_env("DL4DS_STREAM_NO_WS") == nullptr && exp_env("DL4DS_STREAM_NO_WS1") == nullptr;
        if (ws1 && NT <= 3) {
            const float* w0 = sp.c.w;
            bool done = false;
            if (NT == 1) done = launch_stream_ws<1, E, 1, 4>(s, sp, N);
            else if (NT == 2) done = launch_stream_ws<1, E, 2, 4>(s, sp, N);
            else done = launch_stream_ws<1, E, 3, 4>(s, sp, N);
            if (done) return;
            sp.c.w = w0;
        }
    }
    switch (NT) {
        case 1: launch_stream<KS, E, 1, 4>(s, sp, N); break;
        case 2: launch_stream<KS, E, 2, 4>(s, sp, N); break;
        case 3: launch_stream<KS, E, 3, 4>(s, sp, N); break;
        default: launch_stream<KS, E, 4, 4>(s, sp, N); break;
    }
}

}  // namespace

// 5x5 layers (the 9x9 stride-2 transposed convolutions of DeconvolutionBlock as 5x5 convolutions + depth_to_space, deconv.hip):
// only the producer / consumer kernel, 24- or 32-channel chunks
static bool conv2d_stream5_forward(hipStream_t s, const TView& in, const float* w, const TView& out, const ConvEpilogue& ep) {
    static const bool off = exp_env("DL4DS_STREAM_NO_WS") != nullptr || exp_env("DL4DS_STREAM_NO_WS5") != nullptr;
    if (exp_env("DL4DS_STREAM_DEBUG"))
        fprintf(stderr, "stream5: N=%d H=%d W=%d Cin=%d (d2s %d cp %d vec %d) Cout=%d (d2s %d cp %d vec %d) add=%d mask=%d acc=%d\n", in.N, in.H,
                in.W, in.C, in.d2s, in.cp, in.vec, out.C, out.d2s, out.cp, out.vec, ep.add.p != nullptr, ep.mask.p != nullptr, ep.accumulate);
    // 8 input channels (round 5: the ConvLSTM cells' 5x5 input convolutions 8 -> 32 gate channels, which ran on the fallback
    // conv_igemm_kernel at 58 TFLOP/s): one chunk of 8 channels, E = 2.  DL4DS_STREAM5_NO_E2=1 for A/B.
    static const bool no_e2 = exp_env("DL4DS_STREAM5_NO_E2") != nullptr;
    const bool e2 = in.C == 8 && out.C >= 16 && !no_e2;
    if (off || (in.C < 16 && !e2) || (long)in.H * in.W < 256) return false;
    if (!in.vec || !out.vec || (out.C & 3) || (ep.add.p && !ep.add.vec) || (ep.mask.p && !ep.mask.vec)) return false;
    if ((((uintptr_t)ep.bias) & 15) != 0) return false;
    if ((long)cdiv(in.W, 16) * cdiv(in.H, 16) * in.N >= (1l << 20)) return false;
    const int E = e2 ? 2 : ((in.C % 32 == 0) ? 8 : ((in.C % 24 == 0) ? 6 : ((in.C % 16 == 0) ? 4 : 0)));
    if (!E) return false;
    int NT = 0;
    long best = -1;
    // (<= 16 outputs -- the dgrad of a ConvLSTM's 5x5 input kernel, 32 gate channels -> 8: one 16-cout block instead of two,
    //  half the MFMAs; DL4DS_STREAM5_NO_NT1=1 for A/B)
    const int nt_lo = (out.C <= 16 && !exp_env("DL4DS_STREAM5_NO_NT1")) ? 1 : 2;
    for (int pass = 0; pass < 2 && !NT; ++pass)
        for (int nt = nt_lo; nt <= 4; ++nt) {
            // through a depth_to_space store an n-block should not straddle a group (cp channels each); where every choice
            // does (groups of 8 or 16 channels) the kernel's narrow-group form takes the least padded one
            if (pass == 0 && out.d2s > 1 && (16 * nt > out.cp || out.cp % (16 * nt))) continue;
            const long padded = (long)cdiv(out.C, 16 * nt) * 16 * nt;
            if (best < 0 || padded < best || (padded == best && nt > NT)) { best = padded; NT = nt; }
        }
    StreamParams sp;
    ConvParams& p = sp.c;
    p.in = in; p.out = out; p.add = ep.add; p.mask = ep.mask;
    p.w = w; p.bias = ep.bias;
    p.Cin = in.C; p.Cout = out.C; p.H = in.H; p.W = in.W;
    p.relu = ep.relu; p.accumulate = ep.accumulate;
    p.wvec = 0; p.CK = 4 * E; p.TPS = 0;
    if (E == 2) {
        if (NT == 1) NT = 2;
        if (NT == 2) return launch_stream_ws<5, 2, 2, 4>(s, sp, in.N);
        if (NT == 3) return launch_stream_ws<5, 2, 3, 4>(s, sp, in.N);
        return launch_stream_ws<5, 2, 4, 4>(s, sp, in.N);
    }
    if (E == 8) {
        if (NT == 1) return launch_stream_ws<5, 8, 1, 4>(s, sp, in.N);
        if (NT == 2) return launch_stream_ws<5, 8, 2, 4>(s, sp, in.N);
        if (NT == 3) return launch_stream_ws<5, 8, 3, 4>(s, sp, in.N);
        return launch_stream_ws<5, 8, 4, 4>(s, sp, in.N);
    }
    if (E == 6) {
        if (NT == 1) NT = 2;
        if (NT == 2) return launch_stream_ws<5, 6, 2, 4>(s, sp, in.N);
        if (NT == 3) return launch_stream_ws<5, 6, 3, 4>(s, sp, in.N);
        return launch_stream_ws<5, 6, 4, 4>(s, sp, in.N);
    }
    if (NT == 1) return launch_stream_ws<5, 4, 1, 4>(s, sp, in.N);
    if (NT == 2) return launch_stream_ws<5, 4, 2, 4>(s, sp, in.N);
    if (NT == 3) return launch_stream_ws<5, 4, 3, 4>(s, sp, in.N);
    return launch_stream_ws<5, 4, 4, 4>(s, sp, in.N);
}

bool conv2d_stream_forward(hipStream_t s, const TView& in, const float* w, int KS, const TView& out,
                           const ConvEpilogue& ep) {
    if (KS == 5) return conv2d_stream5_forward(s, in, w, out, ep);
    if (KS != 3 && KS != 1) return false;
    // 3x3 with 8 input and >= 16 output channels (the ConvLSTM cells' 3x3 input convolutions 8 -> 32, conv_igemm before round 5):
    // the producer / consumer kernel with one 8-channel chunk
    if (KS == 3 && in.C == 8 && out.C >= 16 && (out.C & 3) == 0 && out.C % 32 == 0 && (long)in.H * in.W >= 256 && in.vec && out.vec &&
        (!ep.add.p || ep.add.vec) && (!ep.mask.p || ep.mask.vec) && ((((uintptr_t)ep.bias) & 15) == 0) &&
        (long)cdiv(in.W, 16) * cdiv(in.H, 16) * in.N < (1l << 20) && !exp_env("DL4DS_STREAM5_NO_E2") && !exp_env("DL4DS_STREAM_NO_WS")) {
        StreamParams sp;
        ConvParams& p = sp.c;
        p.in = in; p.out = out; p.add = ep.add; p.mask = ep.mask;
        p.w = w; p.bias = ep.bias;
        p.Cin = in.C; p.Cout = out.C; p.H = in.H; p.W = in.W;
        p.relu = ep.relu; p.accumulate = ep.accumulate;
        p.wvec = 0; p.CK = 8; p.TPS = 0;
        if (launch_stream_ws<3, 2, 2, 4>(s, sp, in.N)) return true;
    }
    if (in.C < 16 || (long)in.H * in.W < 256) return false;
    if (!in.vec || !out.vec || (out.C & 3) || (ep.add.p && !ep.add.vec) || (ep.mask.p && !ep.mask.vec)) return false;
    if ((((uintptr_t)ep.bias) & 15) != 0) return false;
    if ((long)cdiv(in.W, 16) * cdiv(in.H, 16) * in.N >= (1l << 20)) return false;          // fast_div range
    // cout tiling: NT accumulator tiles per wave, lanes own NT consecutive couts -> needs Cout % NT == 0
    int NT = 0;
    long best = -1;
    static const bool no_ragged = exp_env("DL4DS_STREAM_NO_RAGGED") != nullptr;      // (A/B measurements)
    for (int nt = 1; nt <= 4; ++nt) {
        if ((out.C % nt) && (no_ragged || out.C < 16)) continue;     // Cout % nt != 0: runs on a zero-padded filter copy
        const long padded = (long)cdiv(out.C, 16 * nt) * 16 * nt;
        if (best < 0 || padded < best || (padded == best && nt > NT)) { best = padded; NT = nt; }
    }
    if (out.C < NT) return false;
    // few pixels, many channels (deep U-Net levels: 256->256 at 8x8): the grid is tiles x n-blocks, so narrower
    // n-blocks are the only parallelism there is; each block then streams its own slice of the filter
    {
        const long ntiles = (long)cdiv(in.W, 16) * cdiv(in.H, 16) * in.N;
        while (NT > 1 && ntiles * cdiv(out.C, 16 * NT) < 512) {
            int nt = NT - 1;
            while (nt > 1 && (out.C % nt) && no_ragged) --nt;
            NT = nt;
        }
    }
    // channel chunk: E = CK/4 in {4, 6, 8, 10, 12}, least padded K, then the largest chunk
    int E = 0;
    long bestk = -1;
    for (int e = 4; e <= 12; e += 2) {
        const long padded = (long)cdiv(in.C, 4 * e) * 4 * e;
        if (bestk < 0 || padded < bestk || (padded == bestk && e > E)) { bestk = padded; E = e; }
    }
    // tall tiles (MT = 8): 3x3, cout tiling with NT <= 3 at no extra padding, a grid that still fills the chip, and a
    // chunk width of 16 or 24 channels that does not pad K more than the wide chunk would
    bool tall = false;
    int NT8 = 0, E8 = 0;
    if (KS == 3 && !exp_env("DL4DS_STREAM_NO_TALL")) {
        long bp = -1;
        for (int nt = 1; nt <= 3; ++nt) {
            if ((out.C % nt) && (no_ragged || out.C < 16)) continue;
            const long padded = (long)cdiv(out.C, 16 * nt) * 16 * nt;
            if (bp < 0 || padded < bp || (padded == bp && nt > NT8)) { bp = padded; NT8 = nt; }
        }
        long bk = -1;
        for (int e = 4; e <= 6; e += 2) {
            const long padded = (long)cdiv(in.C, 4 * e) * 4 * e;
            if (bk < 0 || padded < bk || (padded == bk && e > E8)) { bk = padded; E8 = e; }
        }
        const long ntiles8 = (long)cdiv(in.W, 16) * cdiv(in.H, 32) * in.N;
        const bool big = ntiles8 * cdiv(out.C, 16 * NT8) >= 1024 || test_env("DL4DS_STREAM_FORCE_TALL") != nullptr ||
                         test_env("DL4DS_STREAM_FORCE_WS") != nullptr;   // (tests)
        tall = NT8 == 3 && E8 == 6 && bp <= best && bk <= bestk && big;     // (NT 2 / 16-channel chunks measured slower)
        if (exp_env("DL4DS_STREAM_TALL_ANY")) tall = NT8 >= 2 && bp <= best && bk <= bestk && big;     // (experiments)
    }
    StreamParams sp;
    ConvParams& p = sp.c;
    p.in = in; p.out = out; p.add = ep.add; p.mask = ep.mask;
    p.w = w; p.bias = ep.bias;
    p.Cin = in.C; p.Cout = out.C; p.H = in.H; p.W = in.W;
    p.relu = ep.relu; p.accumulate = ep.accumulate;
    p.wvec = 0; p.CK = 4 * E; p.TPS = 0;
    static const bool use_ws = exp_env("DL4DS_STREAM_NO_WS") == nullptr;
    if (tall && use_ws && E8 == 6 && NT8 == 3) {
        p.CK = 24;
        if (launch_stream_ws<3, 6, 3, 8>(s, sp, in.N)) return true;
        p.w = w;                                               // (not eligible: fall through to the two-workgroup kernel)
    }
    if (tall) {
        p.CK = 4 * E8;
        if (E8 == 4) {
            if (NT8 == 2) launch_stream<3, 4, 2, 8>(s, sp, in.N); else launch_stream<3, 4, 3, 8>(s, sp, in.N);
        } else {
            if (NT8 == 2) launch_stream<3, 6, 2, 8>(s, sp, in.N); else launch_stream<3, 6, 3, 8>(s, sp, in.N);
        }
    } else if (KS == 3) {
        switch (E) {
            case 4: dispatch_nt<3, 4>(s, sp, in.N, NT); break;
            case 6: dispatch_nt<3, 6>(s, sp, in.N, NT); break;
            case 8: dispatch_nt<3, 8>(s, sp, in.N, NT); break;
            case 10: dispatch_nt<3, 10>(s, sp, in.N, NT); break;
            default: dispatch_nt<3, 12>(s, sp, in.N, NT); break;
        }
    } else {
        // 1x1 layers are HBM streaming: what matters is that a block's whole tile is in flight at once
        switch (E) {
            case 4: dispatch_nt<1, 4>(s, sp, in.N, NT); break;
            case 6: dispatch_nt<1, 6>(s, sp, in.N, NT); break;
            case 8: dispatch_nt<1, 8>(s, sp, in.N, NT); break;
            case 10: dispatch_nt<1, 10>(s, sp, in.N, NT); break;
            default: dispatch_nt<1, 12>(s, sp, in.N, NT); break;
        }
    }
    return true;
}
