// ConvLSTM2D as ONE launch per layer and direction -- dl4ds/models/blocks.py:339-398 (RecurrentConvBlock: ConvLSTM2D 5x5
// then 3x3, return_sequences=True, tf.keras-2 defaults: tanh / hard_sigmoid, gate order i, f, c, o, zero initial state).
//
// The input convolution of all T frames is one batched launch of the ordinary convolution kernels (graph_ops2.hip).  What
// is left is the recurrence: for every time step a convolution of h_{t-1} with the recurrent kernel, the gate arithmetic,
// and -- backwards -- the same in reverse time.  Run step by step that is 2 T small launches per layer and direction (64^2
// frames x 16 samples = 256 tiles: every launch under-fills the chip, starts cold and pays a kernel boundary).  Here the
// whole sequence runs inside ONE persistent kernel:
//   * a workgroup owns 16 x 16-pixel tiles for all T steps; the recurrent filter sits in LDS, already in MFMA A-fragment
//     order, for the whole launch;
//   * per step: stage the tile + halo of h_{t-1} (backward: of dZ_{t+1}) into LDS, implicit GEMM on v_mfma_f32_16x16x4_f32
//     with the GATE dimension on the MFMA rows, and the gate arithmetic straight on the accumulators: the gate channels are
//     kept interleaved (channel 4 f + gate) in every internal buffer, so the four rows a lane holds are the i, f, g, o
//     pre-activations of ONE (pixel, filter) and z / dz move as float4;
//   * steps are ordered by a flag per tile, not by a grid barrier: tile X may start step t once its 3 x 3 tile
//     neighbourhood has published step t-1.  The hand-off is the guide's form R1, WITHOUT fences: the payload is stored
//     write-through (16-byte sc1 stores), every storing wave drains (s_waitcnt vmcnt(0)), one lane stores the tile's
//     counter (relaxed, agent scope); the consumer polls the counters (relaxed) and stages the halo with sc1 LOADS that
//     bypass its CU's L1 (MI355X_MICROARCH.md "inter-workgroup visibility": per-XCD L2s are not coherent, a CU's L1 is
//     never refreshed by other CUs' stores -- sc1 on both sides is what makes the plain flag sufficient).  Tiles of one
//     image are dealt to one XCD (blockIdx % 8) so that halo exchange stays inside an L2 -- speed only, never correctness.
// Every block of the grid must be resident (a tile waits for its neighbours): grid <= CUs (one 256-thread block per CU,
// LDS-bound; a slice of the CUs is left free while RCCL collectives may run), every spin bounded -- and a spin that gives up sets the
// sticky device error word (runtime.h), which every host-side wait checks and which stops the optimiser kernel: the step fails
// loudly instead of training on stale halos.  The flags are NEVER reset between launches: they are 64-bit words holding
// `epoch + steps done` of a process-wide running epoch (seq_epoch below).  INVARIANT the design rests on: a flag word only ever
// holds 0 (fresh slab, zeroed by Graph::prepare) or a value some earlier launch of THIS process published, i.e. <= the current
// epoch + T.  The words sit at the head of the op's private saved area (graph_ops2.hip: Bufs), which no other op and no other
// kernel of this op writes; a stray larger value would make neighbours read as finished (stale halos, no time-out), so the
// launch wrappers check in debug builds (-DDL4DS_SEQ_CHECK_FLAGS) that no word exceeds the epoch before launching.
#include "ops.h"
#include "prof.h"
#include "head.h"
#include "runtime.h"
#include "dist.h"
#include <algorithm>
#include <vector>

namespace {

int cu_count() {
    static const int n = [] {
        int dev = 0, v = 0;
        HIP_CHECK(hipGetDevice(&dev));
        HIP_CHECK(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
        return v;
    }();
    return n;
}

typedef float f32x4_t __attribute__((ext_vector_type(4)));
using gu32 = __attribute__((address_space(1))) unsigned;
using gu64 = __attribute__((address_space(1))) unsigned long long;

__device__ __forceinline__ float hsig(float z) { return fminf(fmaxf(0.2f * z + 0.5f, 0.f), 1.f); }
__device__ __forceinline__ float dhsig(float z) { return (z >= -2.5f && z <= 2.5f) ? 0.2f : 0.f; }
// tanh through one v_exp_f32 and one v_rcp_f32 (libm's tanhf was ~40 % of the gate arithmetic, which is VALU work squeezed between
// the MFMA phases of a step).  Two regimes, both kept RELATIVELY accurate (round 6):
//   |x| >= 1/4:  dl = 2 / (e^{2|x|} + 1) = 1 - |tanh x|, tanh = sign(x) (1 - dl); the derivative 1 - tanh^2 = dl (2 - dl) is taken from
//                dl, not from the rounded tanh (no cancellation however deep the saturation);
//   |x| <  1/4:  the odd series x (1 - x^2/3 + 2 x^4/15 - 17 x^6/315) (truncation 2e-8 of the value at 1/4).  The first form has an
//                ABSOLUTE error of ~1e-7 -- a staircase of 6e-8 steps around zero.  On cfg4's bench workload (box-blurred positive
//                fields, zero biases) the cell states of the deeper blocks are 1e-7 ... 1e-5: their tanh came out quantised, and the
//                gradients of RecurrentConvBlock3..5, which are proportional to those values, 30 - 75 % from the fp64 oracle (with
//                libm's tanhf in the forward kernel: 1e-6).  TensorFlow's tanh is relatively accurate there (it returns x itself below
//                4e-4), so this was a parity defect, found by the first comparison on data nobody had steered
//                (tests/test_gpu_fullsize.py::test_cfg4_on_the_bench_workload_itself_against_the_oracle, tools/debug/cfg4_bias_probe.py,
//                tools/ubench/tanh_check.hip).  -DSEQ_TANHF / -DSEQ_TANHF_FWD: libm in both kernels / the forward one, for A/B runs.
__device__ __forceinline__ float tanh_small(float x) {
    const float x2 = x * x;
    return x * fmaf(x2, fmaf(x2, fmaf(x2, -17.f / 315.f, 2.f / 15.f), -1.f / 3.f), 1.f);
}
#ifdef SEQ_TANHF_FWD
#define SEQ_TANH_FWD tanhf
#else
#define SEQ_TANH_FWD tanh_fast
#endif
#ifdef SEQ_TANHF
__device__ __forceinline__ float tanh_fast(float x) { return tanhf(x); }
__device__ __forceinline__ float tanh_fast_d(float x, float& d) { const float t = tanhf(x); d = 1.f - t * t; return t; }
#else
__device__ __forceinline__ float tanh_fast(float x) {
    const float ax = fabsf(x);
    const float dl = __fdividef(2.f, __expf(2.f * ax) + 1.f);
    return ax < .25f ? tanh_small(x) : copysignf(1.f - dl, x);
}
__device__ __forceinline__ float tanh_fast_d(float x, float& d) {            // -> tanh x, d = 1 - tanh^2 x
    const float ax = fabsf(x);
    const float dl = __fdividef(2.f, __expf(2.f * ax) + 1.f);
    const float ts = tanh_small(x);
    const bool small = ax < .25f;
    d = small ? fmaf(-ts, ts, 1.f) : dl * (2.f - dl);
    return small ? ts : copysignf(1.f - dl, x);
}
#endif

constexpr unsigned SPIN_LIMIT = 1u << 22;       // ~seconds: a lost neighbour is REPORTED (SeqParams::err), never a hung GPU

// TR = pixel rows per wave: the tile is 16 wide and 4 TR high.  TR = 2 (8 x 16 tiles) doubles the number of tiles: with two
// tiles per workgroup the hand-off latency of one (drain -> flag -> poll -> stage) passes under the other's arithmetic.
template <int KS, int F, bool BWD, int TR>
struct Geom {
    static constexpr int TH = 4 * TR;
    static constexpr int TW = 16 + KS - 1;                  // staged tile: TW wide, TY high
    static constexpr int TY = TH + KS - 1;
    static constexpr int PX = 64 * TR;                      // pixels per tile
    static constexpr int CIN = BWD ? 4 * F : F;             // channels of the staged tensor (dZ' : h)
    static constexpr int NQ = CIN / 4;                      // ... in quads
    static constexpr int RT = BWD ? 1 : F / 4;              // MFMA row tiles: 4F gate rows forward, F (<= 16) rows backward
    // PAIR (round 5, backward with F = 8 and two pixel rows per wave): the 8 filters fill half of the MFMA's 16 rows; rows 8 .. 15
    // take the SAME filters for the pixel ONE ROW BELOW, i.e. the filter displaced by one tap row -- K runs over (KS + 1) x KS
    // taps, and one accumulator holds both pixel rows of the wave: 30 tap steps instead of 2 x 25 at 5 x 5, 12 instead of 18 at 3 x 3
#ifdef DL4DS_SEQ_NO_PAIR
    static constexpr bool PAIR = false;
#else
    static constexpr bool PAIR = BWD && F == 8 && TR == 2;
#endif
    static constexpr int NTY = PAIR ? KS + 1 : KS;          // tap rows of the K loop
    static constexpr int KSTEPS = NTY * KS * NQ;
    static constexpr int W_FLOATS = KSTEPS * RT * 64;
    static constexpr int E = CIN / 4;                       // channels per MFMA k-slot and tap (see seq_chan)
    // floats per staged pixel.  ds_read_b128 is serviced in four NON-contiguous 16-lane groups ({0-3,12-15,20-27}, ...:
    // MI355X_MICROARCH.md, LDS): with lane = (pixel n16, k-slot q) a group mixes eight pixels of one k-slot with the other
    // eight pixels of the next, so the 16 lanes cover all 64 banks iff the k-slots sit ONE 16-byte slot apart and the pixel
    // pitch is 2 (mod 4) slots -- not the odd pitch of the folklore (0.46 of the LDS cycles were conflicts with CIN + 4).
    // 8-byte reads (CIN = 8) are serviced in two 32-lane halves: pitch 12 floats is conflict-free there.
    static constexpr int PITCH = CIN >= 16 ? CIN + 8 : CIN + 4;
    static constexpr int TILE_FLOATS = TY * TW * PITCH;
    static constexpr int XCH_FLOATS = BWD ? 0 : 3 * PX * F;   // forward: h, c, out of a tile on their way to 16-byte stores
    static constexpr size_t LDS_BYTES = (size_t)(W_FLOATS + TILE_FLOATS + XCH_FLOATS) * sizeof(float);
};

struct SeqParams {
    const float* U;        // (KS*KS, F, 4F) recurrent kernel, gate-interleaved columns
    float* Z;              // (B,T,H,W,4F) interleaved: forward in = x-part + bias, out = z; backward in
    float* C;              // (B,T,H,W,F) cell states
    float* Hrec;           // (B,T,H,W,F): frame t = h_{t-1} (frame 0 = 0), written by the forward kernel
    float* out;            // (B,T,H,W,F) layer output ([relu](h_t)); backward: read for the ReLU mask
    const float* dout;     // backward: gradient of out
    float* dZ;             // backward out: (B,T,H,W,4F) interleaved
    float* dc;             // backward scratch: (B,H,W,F) running dL/dc
    unsigned* flags;       // [tiles] 64-bit words: epoch + steps completed by each tile (never reset: see seq_epoch)
    unsigned long long epoch;        // this launch's base value: everything an earlier launch left in `flags` is below it
    unsigned* err;         // host-visible sticky error word (runtime.h): set when a spin gives up
    int B, T, H, W, tiles_x, tiles_y, ntiles, relu, tr;
    unsigned long long* trace;   // DL4DS_SEQ_TRACE: [block][8] phase times (100 MHz wall clock), null otherwise
};

#define SEQ_MARK(ph) do { if (p.trace && threadIdx.x == 0) { const unsigned long long _n = wall_clock64(); p.trace[blockIdx.x * 8 + (ph)] += _n - _t0; _t0 = _n; } } while (0)

// tile index of this block's i-th tile: consecutive tiles (= the tiles of one image) go to blocks of ONE XCD
__device__ __forceinline__ int my_tile(int i, int grid) {
    const int b = blockIdx.x;
    const int per = grid >> 3;
    const int lin = (grid & 7) ? b : (b & 7) * per + (b >> 3);
    return i * grid + lin;
}

typedef int i32x4_t __attribute__((ext_vector_type(4)));
constexpr int RSRC3 = 0x00020000;
constexpr int AUX_SC1 = 16;                     // buffer_load / buffer_store ... sc1: bypass L1 / write through (agent-visible)
constexpr int OOB = (int)0x7ffffff0;            // beyond every descriptor's record count: loads return 0, stores are dropped

// Hand-off between workgroups inside the launch (cdna_hip_programming.md, Guideline 16, form R1): the payload another
// workgroup will read -- h_t forward, dZ_t backward -- is stored WRITE-THROUGH (16-byte sc1 stores), every storing wave
// drains its stores, ONE lane then stores the tile's step counter (relaxed, agent scope).  The consumer polls the counters
// of its 3 x 3 tile neighbourhood (nine lanes, one word each, relaxed) and stages the halo with sc1 LOADS (L2-served, the
// CU's L1 is bypassed): no release / acquire fence on either side, nothing depends on where a workgroup runs.
__device__ __forceinline__ void wait_neighbours(const SeqParams& p, int tile, unsigned need) {
    if (threadIdx.x < 9) {
        const int tpi = p.tiles_x * p.tiles_y;
        const int img = tile / tpi, loc = tile - img * tpi;
        const int ty = loc / p.tiles_x, tx = loc - ty * p.tiles_x;
        const int yy = ty + (int)threadIdx.x / 3 - 1, xx = tx + (int)threadIdx.x % 3 - 1;
        if (threadIdx.x != 4 && yy >= 0 && xx >= 0 && yy < p.tiles_y && xx < p.tiles_x) {
            gu64* f = (gu64*)p.flags + (img * tpi + yy * p.tiles_x + xx);
            unsigned spins = 0;
            const unsigned long long want = p.epoch + need;       // (64-bit running sum: never wraps, a zeroed slab reads as "nothing done")
            while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && ++spins < SPIN_LIMIT)
                __builtin_amdgcn_s_sleep(1);
            // gave up: the halo this tile stages next is stale.  Say so where the host looks after every sync (device_error_check):
            // the step must not pass for a valid one.
            if (spins >= SPIN_LIMIT) __hip_atomic_fetch_or(p.err, DEV_ERR_CONVLSTM_SEQ_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void publish(const SeqParams& p, int tile, unsigned done) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // EVERY storing wave drains its (sc1) stores
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store((gu64*)p.flags + tile, p.epoch + done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// stage the (TY x TW) halo tile of a frame (pixel pitch CIN floats, zero outside the image) as [y][x][Geom::PITCH], sc1 loads
template <int TW, int TY, int CIN>
__device__ __forceinline__ void stage_tile(float* tile, const float* frame, int y0, int x0, int H, int W, int half) {
    constexpr int NQ = CIN / 4;
    constexpr int N = TY * TW * NQ, ITERS = (N + 255) / 256;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(frame), 0, H * W * CIN * 4, RSRC3);
    i32x4_t v[ITERS];
#pragma unroll
    for (int u = 0; u < ITERS; ++u) {
        const int e = threadIdx.x + 256 * u;
        const int cq = e % NQ, pix = e / NQ;
        const int xx = pix % TW, yy = pix / TW;
        const int y = y0 + yy - half, x = x0 + xx - half;
        const bool ok = e < N && y >= 0 && y < H && x >= 0 && x < W;
        v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? ((y * W + x) * CIN + 4 * cq) * 4 : OOB, 0, AUX_SC1);
    }
#pragma unroll
    for (int u = 0; u < ITERS; ++u) {
        const int e = threadIdx.x + 256 * u;
        const int cq = e % NQ, pix = e / NQ;
        const int xx = pix % TW, yy = pix / TW;
        if (e < N) *reinterpret_cast<i32x4_t*>(tile + (size_t)(yy * TW + xx) * (CIN >= 16 ? CIN + 8 : CIN + 4) + 4 * cq) = v[u];
    }
}

// The K loop of both directions.  K is ordered (tap, e) with MFMA k-slot kk owning channels E kk + e, e = 0 .. E - 1, of the
// staged tensor (E = channels / 4): a lane then needs E CONSECUTIVE channels of one pixel per tap -- one or two ds_read_b128
// (ds_read_b64 / b32 for 8 / 4 channels) feed E k-steps, and the filter fragments, stored [tap][row tile][lane][E], come the
// same way: 0.3 LDS reads per MFMA instead of 1.25 (which cost the single wave of a SIMD ~10 issue cycles each).  Software-
// pipelined by hand: the fragments of tap k + 1 are requested before the MFMAs of tap k are issued (two register sets, the
// tap loop unrolled by two) -- fully unrolled the compiler hoists every read and spills at 5 x 5, rolled it waits per tap.
// E consecutive floats from LDS as the widest reads their alignment allows (E * 4 bytes, at most 16)
template <int E>
__device__ __forceinline__ void lds_vec(const float* p, float (&d)[E]) {
    if constexpr (E == 1) {
        d[0] = p[0];
    } else if constexpr (E == 2) {
        const float2 v = *reinterpret_cast<const float2*>(p);
        d[0] = v.x; d[1] = v.y;
    } else {
#pragma unroll
        for (int i = 0; i < E / 4; ++i) {
            const f32x4_t v = *reinterpret_cast<const f32x4_t*>(p + 4 * i);
            d[4 * i] = v[0]; d[4 * i + 1] = v[1]; d[4 * i + 2] = v[2]; d[4 * i + 3] = v[3];
        }
    }
}

// position i of the filter-fragment array -> (tap, row tile a, lane l, k-step e); layout as read by seq_kloop
__host__ __device__ inline void seq_wpos(int E, int RT, int i, int& tap, int& a, int& l, int& e) {
    if (E >= 4) {
        const int w = i & 3; i >>= 2;
        l = i & 63; i >>= 6;
        const int j = i % (E / 4); i /= (E / 4);
        a = i % RT; tap = i / RT;
        e = 4 * j + w;
    } else {
        e = i % E; i /= E;
        l = i & 63; i >>= 6;
        a = i % RT; tap = i / RT;
    }
}
// channel of the staged tensor that MFMA k-slot kk multiplies in k-step e of a tap (E = channels / 4 k-steps per tap)
__host__ __device__ constexpr int seq_chan(int E, int kk, int e) { return E >= 4 ? 16 * (e / 4) + 4 * kk + (e % 4) : E * kk + e; }
// a k-slot's E channels of one staged pixel: E / 4 reads of 16 bytes, 64 bytes apart (E >= 4), else one 8- / 4-byte read
template <int E>
__device__ __forceinline__ void lds_pix(const float* p, float (&d)[E]) {
    if constexpr (E >= 4) {
#pragma unroll
        for (int i = 0; i < E / 4; ++i) {
            const f32x4_t v = *reinterpret_cast<const f32x4_t*>(p + 16 * i);
            d[4 * i] = v[0]; d[4 * i + 1] = v[1]; d[4 * i + 2] = v[2]; d[4 * i + 3] = v[3];
        }
    } else {
        lds_vec<E>(p, d);
    }
}

template <int KS, int CIN, int RT, int TR, int TW, int NTY = KS, int WROWS = TR>     // NTY tap rows; the wave's first pixel row is WROWS * wave
__device__ __forceinline__ void seq_kloop(const float* wA, const float* tile, int lane, int wave, int n16, int q,
                                          f32x4_t (&acc)[RT][TR]) {
    constexpr int E = CIN / 4, P = CIN >= 16 ? CIN + 8 : CIN + 4;
    // k-slot q reads channels seq_chan(q, e): quads 4 q .. 4 q + 3 of every 16-channel group (one 16-byte slot per k-slot)
    const float* bbase = tile + ((size_t)(WROWS * wave) * TW + n16) * P + (E >= 4 ? 4 : E) * q;
    // filter fragments: [tap][row tile][E / 4][lane][4] (E >= 4: every 16-byte read has the lanes side by side), else [tap][row tile][lane][E]
    const float* abase = wA + (size_t)lane * (E >= 4 ? 4 : E);
    float afA[RT][E], bfA[TR][E], afB[RT][E], bfB[TR][E];
    auto load = [&](int tap, int ky, int kx, float (&af)[RT][E], float (&bf)[TR][E]) __attribute__((always_inline)) {
        const float* ap = abase + (size_t)tap * RT * 64 * E;
        const float* bp = bbase + ((size_t)ky * TW + kx) * P;
#pragma unroll
        for (int a = 0; a < RT; ++a) {
            if constexpr (E >= 4) {
#pragma unroll
                for (int j = 0; j < E / 4; ++j) {
                    const f32x4_t v = *reinterpret_cast<const f32x4_t*>(ap + (size_t)(a * (E / 4) + j) * 256);
                    af[a][4 * j] = v[0]; af[a][4 * j + 1] = v[1]; af[a][4 * j + 2] = v[2]; af[a][4 * j + 3] = v[3];
                }
            } else {
                lds_vec<E>(ap + a * 64 * E, af[a]);
            }
        }
#pragma unroll
        for (int r = 0; r < TR; ++r) lds_pix<E>(bp + (size_t)r * TW * P, bf[r]);
    };
    auto mma = [&](const float (&af)[RT][E], const float (&bf)[TR][E]) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int r = 0; r < TR; ++r)
#pragma unroll
                for (int a = 0; a < RT; ++a) acc[a][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a][e], bf[r][e], acc[a][r], 0, 0, 0);
    };
    constexpr int NT = NTY * KS;                 // pairs of taps, then the last one (odd) or two (even) -- no conditional inside the loop
    int ky = 0, kx = 0;
    auto next = [&]() __attribute__((always_inline)) { if (++kx == KS) { kx = 0; ++ky; } };
    load(0, 0, 0, afA, bfA);
#pragma unroll 1
    for (int tap = 0; tap + 2 < NT; tap += 2) {
        next();
        load(tap + 1, ky, kx, afB, bfB);
        mma(afA, bfA);
        next();
        load(tap + 2, ky, kx, afA, bfA);
        mma(afB, bfB);
    }
    if constexpr (NT % 2 == 0) {
        next();
        load(NT - 1, ky, kx, afB, bfB);
        mma(afA, bfA);
        mma(afB, bfB);
    } else {
        mma(afA, bfA);
    }
}

// ================================================================================================ forward
template <int KS, int F, int TR>
__global__ __launch_bounds__(256, 1) void convlstm_seq_fwd_kernel(const SeqParams p) {
    using G = Geom<KS, F, false, TR>;
    constexpr int RT = G::RT, NQ = G::NQ, TW = G::TW, C4 = 4 * F, PX = G::PX;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wA = smem;                              // filter fragments, layout: seq_wpos
    float* tile = smem + G::W_FLOATS;
    float* xch = tile + G::TILE_FLOATS;            // [3][PX pixels][F]: h, c, out of the tile on their way to 16-byte stores
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n16 = lane & 15, q = lane >> 4;
    // recurrent filter -> A fragments: row m of row tile a is gate channel 16 a + m (= filter 4 a + m / 4, gate m % 4)
    for (int i = tid; i < G::W_FLOATS; i += 256) {
        constexpr int E = G::E;
        int tap, a, l, e;
        seq_wpos(E, RT, i, tap, a, l, e);
        wA[i] = p.U[((size_t)tap * F + seq_chan(E, l >> 4, e)) * C4 + 16 * a + (l & 15)];
    }
    __syncthreads();
    const int grid = gridDim.x;
    const bool single = p.ntiles <= grid;          // one tile per workgroup: the cell state never leaves the registers
    const int tpi = p.tiles_x * p.tiles_y;
    const size_t hw = (size_t)p.H * p.W;
    unsigned long long _t0 = p.trace ? wall_clock64() : 0ull;
    float cstate[RT][TR];
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int r = 0; r < TR; ++r) cstate[a][r] = 0.f;
    for (int t = 0; t < p.T; ++t) {
        for (int it = 0;; ++it) {
            const int tl = my_tile(it, grid);
            if (tl >= p.ntiles) { if (it * grid >= p.ntiles) break; else continue; }
            const int img = tl / tpi, loc = tl - img * tpi;
            const int ty = loc / p.tiles_x, tx = loc - ty * p.tiles_x;
            const int y0 = ty * (4 * TR), x0 = tx * 16;
            const size_t fr = ((size_t)img * p.T + t) * hw;               // pixel index of frame (img, t)
            const int x = x0 + n16;
            f32x4_t acc[RT][TR], zx[RT][TR];
            // the input part (+ bias) of z and the previous cell state are requested AFTER the halo has been staged, so that
            // they arrive under the K loop (vector memory returns in order: requested earlier they would stand between the
            // staging loads and their LDS writes)
            auto request = [&]() __attribute__((always_inline)) {
#pragma unroll
                for (int r = 0; r < TR; ++r) {
                    const int y = y0 + TR * wave + r;
                    const bool ok = y < p.H && x < p.W;
                    const size_t pix = fr + (size_t)(ok ? y : 0) * p.W + (ok ? x : 0);
#pragma unroll
                    for (int a = 0; a < RT; ++a) {
                        zx[a][r] = *reinterpret_cast<const f32x4_t*>(p.Z + pix * C4 + 16 * a + 4 * q);
                        if (!single) cstate[a][r] = (t > 0) ? __builtin_nontemporal_load(p.C + (pix - hw) * F + 4 * a + q) : 0.f;
                    }
                }
            };
#pragma unroll
            for (int r = 0; r < TR; ++r)
#pragma unroll
                for (int a = 0; a < RT; ++a) acc[a][r] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            SEQ_MARK(0);
            if (t > 0) {
                wait_neighbours(p, tl, (unsigned)t);
                SEQ_MARK(1);
                stage_tile<TW, G::TY, F>(tile, p.Hrec + fr * F, y0, x0, p.H, p.W, KS / 2);
                __syncthreads();
                SEQ_MARK(2);
                request();
                seq_kloop<KS, F, RT, TR, TW>(wA, tile, lane, wave, n16, q, acc);
            } else {
                request();
            }
            SEQ_MARK(3);
            // gates on the accumulators: (i, f, g, o) of filter 4 a + q at pixel (y, x); z goes out as float4, h / c / out
            // cross the LDS so that they leave as 16-byte stores too (h write-through: the neighbours read it next step)
#pragma unroll
            for (int r = 0; r < TR; ++r) {
                const int y = y0 + TR * wave + r;
                const bool ok = y < p.H && x < p.W;
                const size_t pix = fr + (size_t)(ok ? y : 0) * p.W + (ok ? x : 0);
                const int pl = (TR * wave + r) * 16 + n16;
#pragma unroll
                for (int a = 0; a < RT; ++a) {
                    const f32x4_t z = acc[a][r] + zx[a][r];
                    const float gi = hsig(z[0]), gf = hsig(z[1]), gg = SEQ_TANH_FWD(z[2]), go = hsig(z[3]);
                    const float c = gf * cstate[a][r] + gi * gg;
                    const float h = go * SEQ_TANH_FWD(c);
                    cstate[a][r] = c;
                    const int f = 4 * a + q;
                    if (ok) *reinterpret_cast<f32x4_t*>(p.Z + pix * C4 + 16 * a + 4 * q) = z;
                    xch[pl * F + f] = h;
                    xch[(PX + pl) * F + f] = c;
                    xch[(2 * PX + pl) * F + f] = p.relu ? fmaxf(h, 0.f) : h;
                }
            }
            __syncthreads();
            {
                const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(p.Hrec + (fr + hw) * F, 0, (int)(hw * F * 4), RSRC3);
                for (int e = tid; e < PX * NQ; e += 256) {
                    const int cq = e % NQ, pl = e / NQ;
                    const int y = y0 + (pl >> 4), xx = x0 + (pl & 15);
                    if (y < p.H && xx < p.W) {
                        const size_t pix = fr + (size_t)y * p.W + xx;
                        const i32x4_t hv = *reinterpret_cast<const i32x4_t*>(xch + pl * F + 4 * cq);
                        if (t + 1 < p.T) __builtin_amdgcn_raw_buffer_store_b128(hv, rh, ((y * p.W + xx) * F + 4 * cq) * 4, 0, AUX_SC1);
                        if (t == 0) *reinterpret_cast<f32x4_t*>(p.Hrec + pix * F + 4 * cq) = (f32x4_t){0.f, 0.f, 0.f, 0.f};      // h_{-1} = 0
                        *reinterpret_cast<f32x4_t*>(p.C + pix * F + 4 * cq) = *reinterpret_cast<const f32x4_t*>(xch + (PX + pl) * F + 4 * cq);
                        *reinterpret_cast<f32x4_t*>(p.out + pix * F + 4 * cq) = *reinterpret_cast<const f32x4_t*>(xch + (2 * PX + pl) * F + 4 * cq);
                    }
                }
            }
            SEQ_MARK(4);
            if (t + 1 < p.T) publish(p, tl, (unsigned)(t + 1));
            else __syncthreads();
            SEQ_MARK(5);
        }
    }
}

// ================================================================================================ backward
// dh_{t}[p, ci] (recurrent part) = sum_{tap, c'} dZ'_{t+1}[p - tap + half, c'] U'[tap][ci][c']: a convolution of dZ'_{t+1}
// with the flipped, transposed filter; MFMA rows = the F hidden channels (rows F..15 idle for F < 16), K = taps x 4F.
template <int KS, int F, int TR>
__global__ __launch_bounds__(256, 1) void convlstm_seq_bwd_kernel(const SeqParams p) {
    using G = Geom<KS, F, true, TR>;
    constexpr int NQ = G::NQ, TW = G::TW, C4 = 4 * F;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wA = smem;                              // filter fragments, layout: seq_wpos
    float* tile = smem + G::W_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n16 = lane & 15, q = lane >> 4;
    for (int i = tid; i < G::W_FLOATS; i += 256) {
        constexpr int E = G::E;
        int tap, a_, l, e;
        seq_wpos(E, 1, i, tap, a_, l, e);
        const int ci = l & 15, cc = seq_chan(E, l >> 4, e);
        if constexpr (G::PAIR) {
            // row ci = (half, filter): half 0 = the wave's upper pixel row with tap row ky', half 1 = the row below, whose tap row
            // ky' is tap row ky' - 1 of the filter
            const int kyp = tap / KS, kx = tap - kyp * KS, half = ci >> 3, ky = kyp - half;
            const bool ok = ky >= 0 && ky < KS;
            const int ftap = KS * KS - 1 - ((ok ? ky : 0) * KS + kx);    // flipped tap
            wA[i] = ok ? p.U[((size_t)ftap * F + (ci & 7)) * C4 + cc] : 0.f;
        } else {
            const int ftap = KS * KS - 1 - tap;                          // flipped tap
            wA[i] = (ci < F) ? p.U[((size_t)ftap * F + ci) * C4 + cc] : 0.f;
        }
    }
    __syncthreads();
    const int grid = gridDim.x;
    const int tpi = p.tiles_x * p.tiles_y;
    const size_t hw = (size_t)p.H * p.W;
    unsigned long long _t0 = p.trace ? wall_clock64() : 0ull;
    for (int t = p.T - 1; t >= 0; --t) {
        const unsigned step = (unsigned)(p.T - 1 - t);                   // steps completed before this one
        for (int it = 0;; ++it) {
            const int tl = my_tile(it, grid);
            if (tl >= p.ntiles) { if (it * grid >= p.ntiles) break; else continue; }
            const int img = tl / tpi, loc = tl - img * tpi;
            const int ty = loc / p.tiles_x, tx = loc - ty * p.tiles_x;
            const int y0 = ty * (4 * TR), x0 = tx * 16;
            const size_t fr = ((size_t)img * p.T + t) * hw;
            const int x = x0 + n16;
            constexpr int NACC = G::PAIR ? 1 : TR;                       // PAIR: one accumulator tile holds both pixel rows
            f32x4_t acc[1][NACC];
#pragma unroll
            for (int r = 0; r < NACC; ++r) acc[0][r] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            // everything the gate arithmetic reads is requested right AFTER the halo has been staged, so that it arrives under
            // the K loop (vector memory returns in order: requested before the staging loads it would delay them).  The gate
            // work is spread over ALL lanes: lane (pixel n16, q) takes filters FPL q .. FPL q + FPL - 1 (FPL = F / 4), although
            // the MFMA leaves the recurrent part of dh for filters 4 q' .. 4 q' + 3 in lanes q' < F / 4
            constexpr int FPL = F / 4;
            float dO[TR][FPL], oo[TR][FPL], cc[TR][FPL], cp[TR][FPL], dcn[TR][FPL];
            f32x4_t zz[TR][FPL];
            auto request = [&]() __attribute__((always_inline)) {
#pragma unroll
                for (int r = 0; r < TR; ++r) {
                    const int y = y0 + TR * wave + r;
                    const bool ok = y < p.H && x < p.W;
                    const size_t pix = fr + (size_t)(ok ? y : 0) * p.W + (ok ? x : 0);
                    const size_t sp = ((size_t)img * hw + (size_t)(ok ? y : 0) * p.W + (ok ? x : 0)) * F + FPL * q;
                    __builtin_memcpy(dO[r], p.dout + pix * F + FPL * q, FPL * 4);
                    __builtin_memcpy(oo[r], p.out + pix * F + FPL * q, FPL * 4);
                    __builtin_memcpy(cc[r], p.C + pix * F + FPL * q, FPL * 4);
#pragma unroll
                    for (int i = 0; i < FPL; ++i) { cp[r][i] = 0.f; dcn[r][i] = 0.f; }
                    if (t > 0) __builtin_memcpy(cp[r], p.C + (pix - hw) * F + FPL * q, FPL * 4);
                    if (t + 1 < p.T) __builtin_memcpy(dcn[r], p.dc + sp, FPL * 4);
#pragma unroll
                    for (int i = 0; i < FPL; ++i) zz[r][i] = *reinterpret_cast<const f32x4_t*>(p.Z + pix * C4 + 4 * (FPL * q + i));
                }
            };
            SEQ_MARK(0);
            if (t + 1 < p.T) {
                wait_neighbours(p, tl, step);
                SEQ_MARK(1);
                stage_tile<TW, G::TY, C4>(tile, p.dZ + (fr + hw) * C4, y0, x0, p.H, p.W, KS / 2);
                __syncthreads();
                SEQ_MARK(2);
                request();
                seq_kloop<KS, C4, 1, NACC, TW, G::NTY, TR>(wA, tile, lane, wave, n16, q, acc);
            } else {
                request();
            }
            SEQ_MARK(3);
            // gate backward; dZ_t is what the neighbours stage next step: write-through 16-byte stores
            const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(p.dZ + fr * C4, 0, (int)(hw * C4 * 4), RSRC3);
            const int src_lane = n16 + 16 * ((FPL * q) >> 2), j0 = (FPL * q) & 3;
#pragma unroll
            for (int r = 0; r < TR; ++r) {
                const int y = y0 + TR * wave + r;
                // the recurrent part of dh for this lane's filters sits in lane src_lane, accumulator rows j0 .. j0 + FPL - 1
                float rec[4];
                if constexpr (G::PAIR) {
                    // (rows 4 q' .. 4 q' + 3 of the one accumulator: q' = 0, 1 the upper pixel row's filters, q' = 2, 3 the lower one's)
#pragma unroll
                    for (int j = 0; j < 4; ++j) rec[j] = __shfl(acc[0][0][j], src_lane + 32 * r, 64);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) rec[j] = (FPL == 4) ? acc[0][r][j] : __shfl(acc[0][r][j], src_lane, 64);
                }
                if (y < p.H && x < p.W) {
                    const size_t sp = ((size_t)img * hw + (size_t)y * p.W + x) * F + FPL * q;      // per-sample state (dc)
                    float dcout[FPL];
#pragma unroll
                    for (int i = 0; i < FPL; ++i) {
                        const f32x4_t z = zz[r][i];
                        float dgg, dtc;
                        const float gi = hsig(z[0]), gf = hsig(z[1]), gg = tanh_fast_d(z[2], dgg), go = hsig(z[3]);
                        float dh = dO[r][i];
                        if (p.relu) dh = (oo[r][i] > 0.f) ? dh : 0.f;
                        float rc = rec[i];
                        if (FPL == 2) rc = j0 ? rec[2 + i] : rec[i];
                        if (FPL == 1) rc = j0 == 0 ? rec[0] : (j0 == 1 ? rec[1] : (j0 == 2 ? rec[2] : rec[3]));
                        dh += rc;
                        const float tc = tanh_fast_d(cc[r][i], dtc);
                        const float dc = dh * go * dtc + dcn[r][i];
                        f32x4_t dz;
                        dz[0] = dc * gg * dhsig(z[0]);
                        dz[1] = dc * cp[r][i] * dhsig(z[1]);
                        dz[2] = dc * gi * dgg;
                        dz[3] = dh * tc * dhsig(z[3]);
                        i32x4_t dzi;
                        __builtin_memcpy(&dzi, &dz, 16);
                        __builtin_amdgcn_raw_buffer_store_b128(dzi, rz, (((y * p.W + x) * C4) + 4 * (FPL * q + i)) * 4, 0, AUX_SC1);
                        dcout[i] = dc * gf;
                    }
                    if (t > 0) __builtin_memcpy(p.dc + sp, dcout, FPL * 4);
                }
            }
            SEQ_MARK(4);
            if (t > 0) publish(p, tl, step + 1);
            else __syncthreads();
            SEQ_MARK(5);
        }
    }
}

// column permutation between the Keras gate-major layout (column = gate * F + f) and the interleaved one (4 f + gate); up to
// three arrays per launch (kernel, recurrent kernel, bias of one layer: blockIdx.y picks the job)
struct GateJobs { const float* src[3]; float* dst[3]; int rows[3]; int acc[3]; };
__global__ void gate_interleave_kernel(const GateJobs jobs, int F, int to_interleaved) {
    const int j = blockIdx.y;
    const float* __restrict__ src = jobs.src[j];
    float* __restrict__ dst = jobs.dst[j];
    const int rows = jobs.rows[j], accumulate = jobs.acc[j];
    const int C4 = 4 * F;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < rows * C4; e += gridDim.x * blockDim.x) {
        const int row = e / C4, c = e - row * C4;
        // e indexes the DESTINATION; c is its column
        int sc;
        if (to_interleaved) { const int f = c >> 2, gate = c & 3; sc = gate * F + f; }
        else { const int gate = c / F, f = c - gate * F; sc = 4 * f + gate; }
        const float v = src[(size_t)row * C4 + sc];
        dst[e] = accumulate ? dst[e] + v : v;
    }
}

template <int KS, int F, int TR>
void launch_fwd(hipStream_t s, const SeqParams& p, int grid) {
    using G = Geom<KS, F, false, TR>;
    static bool once = false;
    if (!once) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(convlstm_seq_fwd_kernel<KS, F, TR>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES));
        once = true;
    }
    DL4DS_LAUNCH((convlstm_seq_fwd_kernel<KS, F, TR>), dim3(grid), dim3(256), G::LDS_BYTES, s, p);
    HIP_CHECK(hipGetLastError());
}
template <int KS, int F, int TR>
void launch_bwd(hipStream_t s, const SeqParams& p, int grid) {
    using G = Geom<KS, F, true, TR>;
    static bool once = false;
    if (!once) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(convlstm_seq_bwd_kernel<KS, F, TR>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES));
        once = true;
    }
    DL4DS_LAUNCH((convlstm_seq_bwd_kernel<KS, F, TR>), dim3(grid), dim3(256), G::LDS_BYTES, s, p);
    HIP_CHECK(hipGetLastError());
}

}  // namespace

bool convlstm_seq_supported(int KS, int F, int H, int W, int B) {
    if (getenv("DL4DS_NO_CONVLSTM_SEQ")) return false;
    // weight-gradient kernels on the second compute stream (opt-in, DL4DS_AUX_STREAM=1) may hold CUs for the whole backward pass:
    // residency of a 256-workgroup launch cannot be promised next to them -> the step-by-step path
    if (getenv("DL4DS_AUX_STREAM")) return false;
    if (!(KS == 3 || KS == 5) || !(F == 4 || F == 8 || F == 16)) return false;
    if (F == 16 && KS == 5) return false;                  // backward: filter fragments + the dZ halo tile exceed the LDS
    const long tiles = (long)cdiv(H, 16) * cdiv(W, 16) * B;
    return tiles >= 1 && tiles < (1l << 24);
}

void convlstm_gate_interleave_n(hipStream_t s, int njobs, const float* const* src, float* const* dst, const int* rows, const int* accumulate,
                                int F, bool to_interleaved) {
    GateJobs jobs{};
    int nmax = 0;
    for (int j = 0; j < njobs && j < 3; ++j) {
        jobs.src[j] = src[j]; jobs.dst[j] = dst[j]; jobs.rows[j] = rows[j]; jobs.acc[j] = accumulate[j];
        nmax = std::max(nmax, rows[j] * 4 * F);
    }
    if (njobs <= 0 || nmax == 0) return;
    DL4DS_LAUNCH(gate_interleave_kernel, dim3(std::min(cdiv(nmax, 256), 512), njobs), dim3(256), 0, s, jobs, F, to_interleaved ? 1 : 0);
    HIP_CHECK(hipGetLastError());
}
void convlstm_gate_interleave(hipStream_t s, const float* src, float* dst, int rows, int F, bool to_interleaved, bool accumulate) {
    const int acc = accumulate ? 1 : 0;
    convlstm_gate_interleave_n(s, 1, &src, &dst, &rows, &acc, F, to_interleaved);
}

// DL4DS_SEQ_TRACE=1 (development): per-phase times of every launch on stderr (synchronises)
static unsigned long long* trace_begin(SeqParams& p, int grid, hipStream_t s) {
    static const bool on = exp_env("DL4DS_SEQ_TRACE") != nullptr;
    if (!on) return nullptr;
    static unsigned long long* buf = nullptr;
    if (!buf) HIP_CHECK(hipMalloc((void**)&buf, 4096 * 8 * sizeof(unsigned long long)));
    HIP_CHECK(hipMemsetAsync(buf, 0, (size_t)grid * 8 * sizeof(unsigned long long), s));
    p.trace = buf;
    return buf;
}
static void trace_end(const char* what, const SeqParams& p, int grid, hipStream_t s) {
    if (!p.trace) return;
    HIP_CHECK(hipStreamSynchronize(s));
    std::vector<unsigned long long> h((size_t)grid * 8);
    HIP_CHECK(hipMemcpy(h.data(), p.trace, h.size() * 8, hipMemcpyDeviceToHost));
    double ph[8] = {0};
    for (int b = 0; b < grid; ++b) for (int k = 0; k < 8; ++k) ph[k] += (double)h[(size_t)b * 8 + k] / grid * 0.01;      // us
    fprintf(stderr, "%s T=%d tiles=%d grid=%d tr=%d: prefetch %.1f wait %.1f stage %.1f kloop %.1f gates+stores %.1f publish %.1f us (mean per block, whole launch)\n",
            what, p.T, p.ntiles, grid, p.tr, ph[0], ph[1], ph[2], ph[3], ph[4], ph[5]);
}
// Every workgroup of the launch must be resident at once (one per CU: LDS-bound).  While RCCL collectives of the bucketed
// backward may be running, a slice of the CUs is left to them, so that their workgroups never have to queue behind -- or hold
// up -- a workgroup of this kernel (DL4DS_SEQ_RESERVE_CUS overrides the 32).  Co-residency lost anyway (a late peer rank keeps a
// collective's kernel on the chip for seconds) ends in the error word, not in wrong numbers.
static int seq_grid(const SeqParams& p) {
    static const int reserve = [] { const char* e = getenv("DL4DS_SEQ_RESERVE_CUS"); return e ? atoi(e) : 32; }();
    const int cus = std::max(cu_count() - (dist_active() ? reserve : 0), 8);
    return std::min(p.ntiles, cus);
}

// rows per wave.  Backward: 8 x 16 tiles while 16 x 16 tiles would give a workgroup fewer than two of them (see Geom; its
// step ends with the drain of 32 KB of write-through dZ stores, which the second tile's arithmetic covers).  Forward: 16 x 16
// (its hand-off is 8 KB and the smaller tiles' extra halo costs more than it hides).  Measured at 16 x 8 x 64^2, F = 8:
// backward 5x5 184 -> 173 us, 3x3 123 -> 109 us per launch with two 8 x 16 tiles; forward 87 / 62 us with 16 x 16, 95 / 68 with 8 x 16.
static int seq_tr(int H, int W, int B, bool backward) {
    if (const char* e = exp_env("DL4DS_CONVLSTM_SEQ_TR")) return atoi(e) == 2 ? 2 : 4;
    const long t16 = (long)cdiv(H, 16) * cdiv(W, 16) * B;
    return (backward && t16 < 2l * std::max(cu_count(), 8) && H > 8) ? 2 : 4;
}

// Flags are never zeroed between launches (two memset nodes per layer and step were 30 dispatches of a cfg4 step): a launch
// publishes `epoch + steps done` (1 <= steps <= T) and waits for `epoch + needed`; the epoch is a process-wide running sum of
// T + 1 per launch (64 bits: a 32-bit sum would pass 2^31 after a day and a half of cfg4 steps, and a freshly zeroed slab
// would then compare as ahead), so whatever an earlier launch -- of any layer, tiling or direction -- left in a flag word is <= the new
// epoch and reads as "nothing done yet".  A freshly allocated slab is zeroed (Graph::prepare).
static unsigned long long seq_epoch(int T) {
    static unsigned long long next = 1;
    const unsigned long long e = next;
    next += (unsigned long long)T + 1;
    return e;
}

static SeqParams seq_params(const float* U, float* Z, float* C, float* Hrec, float* out, const float* dout, float* dZ, float* dc,
                            unsigned* flags, int B, int T, int H, int W, int relu, bool backward) {
    SeqParams p;
    p.U = U; p.Z = Z; p.C = C; p.Hrec = Hrec; p.out = out; p.dout = dout; p.dZ = dZ; p.dc = dc; p.flags = flags;
    p.B = B; p.T = T; p.H = H; p.W = W; p.relu = relu;
    p.err = device_error_word();
    p.tr = seq_tr(H, W, B, backward);
    p.trace = nullptr;
    p.epoch = seq_epoch(T);
    p.tiles_x = cdiv(W, 16); p.tiles_y = cdiv(H, 4 * p.tr); p.ntiles = p.tiles_x * p.tiles_y * B;
#ifdef DL4DS_SEQ_CHECK_FLAGS
    {   // the invariant of the header comment, checked the slow way (drains the device: debug builds only)
        std::vector<unsigned long long> h((size_t)p.ntiles);
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemcpy(h.data(), flags, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (unsigned long long v : h) DL4DS_REQUIRE(v < p.epoch, "convlstm_seq: a tile flag is ahead of the epoch (stray write into the flag area)");
    }
#endif
    return p;
}

size_t convlstm_seq_flag_bytes(int H, int W, int B) { return (size_t)cdiv(H, 8) * cdiv(W, 16) * B * sizeof(unsigned long long); }

void convlstm_seq_forward(hipStream_t s, const float* U_il, float* Z_il, float* C, float* Hrec, float* out, unsigned* flags,
                          int B, int T, int H, int W, int KS, int F, int relu) {
    SeqParams p = seq_params(U_il, Z_il, C, Hrec, out, nullptr, nullptr, nullptr, flags, B, T, H, W, relu, false);
    const double px = (double)B * T * H * W;
    ProfScope ps(s, "convlstm_seq_fwd<" + std::to_string(KS) + "," + std::to_string(F) + ">",
                 2.0 * (double)B * (T - 1) * H * W * KS * KS * F * 4 * F, 4.0 * px * (4 * F * 2 + F * 4));
    const int grid = seq_grid(p);
    trace_begin(p, grid, s);
#define DL4DS_SEQ_CASE(K_, F_) if (KS == K_ && F == F_) { if (p.tr == 2) launch_fwd<K_, F_, 2>(s, p, grid); else launch_fwd<K_, F_, 4>(s, p, grid); trace_end("convlstm_seq_fwd", p, grid, s); return; }
    DL4DS_SEQ_CASE(3, 4) DL4DS_SEQ_CASE(3, 8) DL4DS_SEQ_CASE(3, 16) DL4DS_SEQ_CASE(5, 4) DL4DS_SEQ_CASE(5, 8)
#undef DL4DS_SEQ_CASE
    DL4DS_REQUIRE(false, "convlstm_seq_forward: unsupported (KS, F)");
}

void convlstm_seq_backward(hipStream_t s, const float* U_il, const float* Z_il, const float* C, const float* out, const float* dout,
                           float* dZ_il, float* dc, unsigned* flags, int B, int T, int H, int W, int KS, int F, int relu) {
    SeqParams p = seq_params(U_il, const_cast<float*>(Z_il), const_cast<float*>(C), nullptr, const_cast<float*>(out), dout, dZ_il, dc,
                             flags, B, T, H, W, relu, true);
    const double px = (double)B * T * H * W;
    ProfScope ps(s, "convlstm_seq_bwd<" + std::to_string(KS) + "," + std::to_string(F) + ">",
                 2.0 * (double)B * (T - 1) * H * W * KS * KS * F * 4 * F, 4.0 * px * (4 * F * 2 + F * 6));
    const int grid = seq_grid(p);
    trace_begin(p, grid, s);
#define DL4DS_SEQ_CASE(K_, F_) if (KS == K_ && F == F_) { if (p.tr == 2) launch_bwd<K_, F_, 2>(s, p, grid); else launch_bwd<K_, F_, 4>(s, p, grid); trace_end("convlstm_seq_bwd", p, grid, s); return; }
    DL4DS_SEQ_CASE(3, 4) DL4DS_SEQ_CASE(3, 8) DL4DS_SEQ_CASE(3, 16) DL4DS_SEQ_CASE(5, 4) DL4DS_SEQ_CASE(5, 8)
#undef DL4DS_SEQ_CASE
    DL4DS_REQUIRE(false, "convlstm_seq_backward: unsupported (KS, F)");
}
