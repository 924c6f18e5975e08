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
        hipLaunchKernelGGL(pad_filter_kernel, dim3(std::min(cdiv(rows * sp.cw, 256), 1024)), dim3(256), 0, s, p.w, wp, rows, p.Cout, sp.cw);
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
                 2.0 * px * KS * KS * p.Cin * p.Cout, 4.0 * (px * (p.Cin + p.Cout) + (double)KS * KS * p.Cin * p.Cout));
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
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kStreamThreads), lds, s, sp);
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

template <int KS, int E>
void dispatch_nt(hipStream_t s, StreamParams& sp, int N, int NT) {
    switch (NT) {
        case 1: launch_stream<KS, E, 1, 4>(s, sp, N); break;
        case 2: launch_stream<KS, E, 2, 4>(s, sp, N); break;
        case 3: launch_stream<KS, E, 3, 4>(s, sp, N); break;
        default: launch_stream<KS, E, 4, 4>(s, sp, N); break;
    }
}

}  // namespace

bool conv2d_stream_forward(hipStream_t s, const TView& in, const float* w, int KS, const TView& out,
                           const ConvEpilogue& ep) {
    if (KS != 3 && KS != 1) return false;
    if (in.C < 16 || (long)in.H * in.W < 256) return false;
    if (!in.vec || !out.vec || (out.C & 3) || (ep.add.p && !ep.add.vec) || (ep.mask.p && !ep.mask.vec)) return false;
    if ((((uintptr_t)ep.bias) & 15) != 0) return false;
    if ((long)cdiv(in.W, 16) * cdiv(in.H, 16) * in.N >= (1l << 20)) return false;          // fast_div range
    // cout tiling: NT accumulator tiles per wave, lanes own NT consecutive couts -> needs Cout % NT == 0
    int NT = 0;
    long best = -1;
    static const bool no_ragged = getenv("DL4DS_STREAM_NO_RAGGED") != nullptr;      // (A/B measurements)
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
    if (KS == 3 && !getenv("DL4DS_STREAM_NO_TALL")) {
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
        const bool big = ntiles8 * cdiv(out.C, 16 * NT8) >= 1024 || getenv("DL4DS_STREAM_FORCE_TALL") != nullptr;   // (tests)
        tall = NT8 == 3 && E8 == 6 && bp <= best && bk <= bestk && big;     // (NT 2 / 16-channel chunks measured slower)
        if (getenv("DL4DS_STREAM_TALL_ANY")) tall = NT8 >= 2 && bp <= best && bk <= bestk && big;     // (experiments)
    }
    StreamParams sp;
    ConvParams& p = sp.c;
    p.in = in; p.out = out; p.add = ep.add; p.mask = ep.mask;
    p.w = w; p.bias = ep.bias;
    p.Cin = in.C; p.Cout = out.C; p.H = in.H; p.W = in.W;
    p.relu = ep.relu; p.accumulate = ep.accumulate;
    p.wvec = 0; p.CK = 4 * E; p.TPS = 0;
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
