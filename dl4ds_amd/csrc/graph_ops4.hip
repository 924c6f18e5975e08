// The tail of the spatio-temporal nets with HR auxiliary channels AND a LocalizedConvBlock, as ONE pass per direction.
//
// Reference (dl4ds/models/spt_postups.py:133-151, spt_preups.py:114-132; blocks.py:301-333):
//     s    = repeat(expand_dims(ConvBlock_aux(s_in), 1), T)                # (B,T,H,W,CS): T copies
//     x24  = Concatenate([x, s])                                           # (B,T,H,W,CX+CS)
//     lws  = LocallyConnected2D_1x1(relu(Conv1x1(x24; Wt, bt)); Wl, bl)    # LocalizedConvBlock: (B,T,H,W,2)
//     x26  = Concatenate([x24, lws])
//     y    = relu(Conv1x1(x26; W, b))                                      # TransitionLast -> CO channels
// Run layer by layer that is five kernels and ~560 bytes of traffic per HR pixel forward (the time repeat writes T copies of
// what exists once, both concatenations are materialised, every producer stores PART of a 24- / 26-channel pixel: cfg4 spent
// 1.45 ms of 12.4 in concat_join / concat_split / repeat_time alone).  Both 1x1 convolutions are linear in the channel
// segments of their input, and the auxiliary segment does not depend on t:
//     u    = relu(Wt_x x + [Wt_s s + bt])              lws = Wl u + bl              y = relu(W_x x + [W_s s + b] + W_l lws)
// so a thread takes one (sample, grid point), computes the bracketed terms ONCE and walks the T frames: 64 B read + 52 B
// written per pixel (+ 16 B saved for the backward pass).  The backward pass is the same walk: dx, ds (summed over t in
// registers), dWl / dbl (per sample, summed by a small kernel), and ONE (CO + 2)-channel tensor dz = [dy masked ; du masked]
// whose 1x1 weight gradients against x, s (time-summed dz) and lws are three calls of the ordinary weight-gradient kernels --
// they yield dW, db, dWt, dbt at once.  Same variables as the reference's layers (names unchanged); the model tests compare
// with the oracle's layer-by-layer evaluation, DL4DS_NO_REC_TAIL_FUSION=1 builds the separate layers.
#include "graph.h"
#include "prof.h"
#include <algorithm>
#include <cstdlib>

namespace {

inline bool wants_grad(const Graph& g, int tid, const BwdCtx& c) {
    const GTensor& t = g.tensors[tid];
    return t.requires_grad && (!t.is_input || c.input_grads) && (c.param_grads || t.dep_grad_input || exp_env("DL4DS_NO_BWD_PRUNE") != nullptr);
}

typedef int i32x4_t __attribute__((ext_vector_type(4)));
constexpr int RSRC3 = 0x00020000;

// The filter arrays are read with wave-uniform indices.  Through a generic pointer the compiler may not assume that the stores
// of the pass leave them alone: it re-read them with VECTOR loads after every store (280 global loads per frame in the backward
// pass).  Read through the constant address space they are scalar loads, invariant for the launch.
typedef const float __attribute__((address_space(4))) * cfloat_p;
__device__ __forceinline__ cfloat_p uniform_ro(const float* p) { return (cfloat_p)p; }

// (inside the frame loop: the filter pointer made opaque per frame, so that the scalar loads stay next to their uses instead of
//  being hoisted out of the loop and spilled -- 200+ scalar registers parked in vector lanes, a v_readlane per use)
#define REC_TAIL_OPAQUE(p_) asm volatile("" : "+s"(p_))

struct TailParams {
    const float *x, *s, *wt, *bt, *wl, *bl, *w, *b;
    float *y, *u, *lws;
    int B, T, HW;
};

// CO floats at a 4-byte aligned address (pixel pitch CO floats): 16-byte buffer accesses only need dword alignment
template <int CO>
__device__ __forceinline__ void store_row(const __amdgpu_buffer_rsrc_t r, int byte_off, const float (&v)[CO]) {
#pragma unroll
    for (int q = 0; q + 4 <= CO; q += 4) {
        const f32x4 t = {v[q], v[q + 1], v[q + 2], v[q + 3]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, t), r, byte_off + 4 * q, 0, 0);
    }
#pragma unroll
    for (int q = CO & ~3; q < CO; ++q) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v[q]), r, byte_off + 4 * q, 0, 0);
}
template <int CO>
__device__ __forceinline__ void load_row(const __amdgpu_buffer_rsrc_t r, int byte_off, float (&v)[CO]) {
#pragma unroll
    for (int q = 0; q + 4 <= CO; q += 4) {
        const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off + 4 * q, 0, 0));
        v[q] = t[0]; v[q + 1] = t[1]; v[q + 2] = t[2]; v[q + 3] = t[3];
    }
#pragma unroll
    for (int q = CO & ~3; q < CO; ++q) v[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off + 4 * q, 0, 0));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p)), 0, 0x7ffffff0, RSRC3);
}

// thread = (sample, grid point); frames of a sample are (T HW) pixels apart.  The weights are read with wave-uniform
// indices: scalar loads, used straight as FMA operands.
template <int CX, int CS, int CO>
__global__ void __launch_bounds__(256) rec_tail_fwd_kernel(const TailParams a) {
    cfloat_p wt_ = uniform_ro(a.wt), bt_ = uniform_ro(a.bt), w_ = uniform_ro(a.w), b_ = uniform_ro(a.b);
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= (size_t)a.B * a.HW) return;
    const int bi = (int)(p / a.HW), hw = (int)(p - (size_t)bi * a.HW);
    float sv[CS];
#pragma unroll
    for (int q = 0; q < CS; q += 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(a.s + p * CS + q);
        sv[q] = t[0]; sv[q + 1] = t[1]; sv[q + 2] = t[2]; sv[q + 3] = t[3];
    }
    float su[2] = {bt_[0], bt_[1]}, sy[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) sy[o] = b_[o];
#pragma unroll
    for (int j = 0; j < CS; ++j) {
        su[0] = fmaf(sv[j], wt_[(CX + j) * 2], su[0]);
        su[1] = fmaf(sv[j], wt_[(CX + j) * 2 + 1], su[1]);
#pragma unroll
        for (int o = 0; o < CO; ++o) sy[o] = fmaf(sv[j], w_[(CX + j) * CO + o], sy[o]);
    }
    const f32x4 wl4 = *reinterpret_cast<const f32x4*>(a.wl + (size_t)hw * 4);         // Wl[h][w][c][f] -> c * 2 + f
    const float bl0 = a.bl[(size_t)hw * 2], bl1 = a.bl[(size_t)hw * 2 + 1];
    // wave-uniform descriptor based at the first frame of the BLOCK's first sample: the per-lane byte offsets then span at most the
    // few samples a block of 256 grid points touches (< 2^31 checked by the host per sample, whatever the batch size)
    const size_t fbase = (size_t)(((size_t)blockIdx.x * 256) / a.HW) * a.T * a.HW;
    const __amdgpu_buffer_rsrc_t ry = rsrc_of(a.y + fbase * CO);
    for (int t = 0; t < a.T; ++t) {
        const size_t px = ((size_t)bi * a.T + t) * a.HW + hw;
        float xv[CX];
#pragma unroll
        for (int q = 0; q < CX; q += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(a.x + px * CX + q);
            xv[q] = v[0]; xv[q + 1] = v[1]; xv[q + 2] = v[2]; xv[q + 3] = v[3];
        }
        float u0 = su[0], u1 = su[1];
#pragma unroll
        for (int i = 0; i < CX; ++i) { u0 = fmaf(xv[i], wt_[i * 2], u0); u1 = fmaf(xv[i], wt_[i * 2 + 1], u1); }
        u0 = fmaxf(u0, 0.f); u1 = fmaxf(u1, 0.f);
        const float l0 = fmaf(u1, wl4[2], fmaf(u0, wl4[0], bl0)), l1 = fmaf(u1, wl4[3], fmaf(u0, wl4[1], bl1));
        float yv[CO];
#pragma unroll
        for (int o = 0; o < CO; ++o) yv[o] = fmaf(l1, w_[(CX + CS + 1) * CO + o], fmaf(l0, w_[(CX + CS) * CO + o], sy[o]));
#pragma unroll
        for (int i = 0; i < CX; ++i)
#pragma unroll
            for (int o = 0; o < CO; ++o) yv[o] = fmaf(xv[i], w_[i * CO + o], yv[o]);
#pragma unroll
        for (int o = 0; o < CO; ++o) yv[o] = fmaxf(yv[o], 0.f);
        store_row<CO>(ry, (int)((px - fbase) * CO * 4), yv);
        *reinterpret_cast<float2*>(a.u + px * 2) = make_float2(u0, u1);
        *reinterpret_cast<float2*>(a.lws + px * 2) = make_float2(l0, l1);
    }
}

// ---- the same walk with every global access coalesced: a block = 256 consecutive grid points of one sample (HW % 256 == 0);
// per frame the block's x rows arrive as one contiguous float4 stream into LDS ([pixel][CX + 4]: conflict-free 16-byte row
// reads), the y rows leave through LDS as one contiguous stream (a 13-float row is 52 bytes: row-per-lane stores touched four
// 64-byte segments per lane quad and ran at 2.4 TB/s).  The next frame's loads are in flight during the arithmetic.
template <int CX, int CS, int CO>
__global__ void __launch_bounds__(256) rec_tail_fwd_staged_kernel(const TailParams a) {
    cfloat_p wt_ = uniform_ro(a.wt), bt_ = uniform_ro(a.bt), w_ = uniform_ro(a.w), b_ = uniform_ro(a.b);
    constexpr int XQ = CX / 4, XP = CX + 4, NY4 = 256 * CO / 4, YIT = (NY4 + 255) / 256;
    __shared__ __attribute__((aligned(16))) float xs[256 * XP];
    __shared__ __attribute__((aligned(16))) float ys[256 * CO];
    const int tid = threadIdx.x;
    const int blocks_per_sample = a.HW / 256;
    const int bi = blockIdx.x / blocks_per_sample, hw0 = (blockIdx.x - bi * blocks_per_sample) * 256, hw = hw0 + tid;
    const size_t p = (size_t)bi * a.HW + hw;
    float sv[CS];
#pragma unroll
    for (int q = 0; q < CS; q += 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(a.s + p * CS + q);
        sv[q] = t[0]; sv[q + 1] = t[1]; sv[q + 2] = t[2]; sv[q + 3] = t[3];
    }
    float su[2] = {bt_[0], bt_[1]}, sy[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) sy[o] = b_[o];
#pragma unroll
    for (int j = 0; j < CS; ++j) {
        su[0] = fmaf(sv[j], wt_[(CX + j) * 2], su[0]);
        su[1] = fmaf(sv[j], wt_[(CX + j) * 2 + 1], su[1]);
#pragma unroll
        for (int o = 0; o < CO; ++o) sy[o] = fmaf(sv[j], w_[(CX + j) * CO + o], sy[o]);
    }
    const f32x4 wl4 = *reinterpret_cast<const f32x4*>(a.wl + (size_t)hw * 4);
    const float bl0 = a.bl[(size_t)hw * 2], bl1 = a.bl[(size_t)hw * 2 + 1];
    f32x4 xin[XQ];
    auto fetch = [&](int t) __attribute__((always_inline)) {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.x + (((size_t)bi * a.T + t) * a.HW + hw0) * CX);
#pragma unroll
        for (int k = 0; k < XQ; ++k) xin[k] = src[k * 256 + tid];
    };
    fetch(0);
    for (int t = 0; t < a.T; ++t) {
        const size_t px0 = ((size_t)bi * a.T + t) * a.HW + hw0;
#pragma unroll
        for (int k = 0; k < XQ; ++k) {
            const int e = k * 256 + tid;
            *reinterpret_cast<f32x4*>(xs + (e / XQ) * XP + (e % XQ) * 4) = xin[k];
        }
        __syncthreads();
        if (t + 1 < a.T) fetch(t + 1);
        REC_TAIL_OPAQUE(wt_); REC_TAIL_OPAQUE(w_);
        float xv[CX];
#pragma unroll
        for (int q = 0; q < CX; q += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xs + tid * XP + q);
            xv[q] = v[0]; xv[q + 1] = v[1]; xv[q + 2] = v[2]; xv[q + 3] = v[3];
        }
        float u0 = su[0], u1 = su[1];
#pragma unroll
        for (int i = 0; i < CX; ++i) { u0 = fmaf(xv[i], wt_[i * 2], u0); u1 = fmaf(xv[i], wt_[i * 2 + 1], u1); }
        u0 = fmaxf(u0, 0.f); u1 = fmaxf(u1, 0.f);
        const float l0 = fmaf(u1, wl4[2], fmaf(u0, wl4[0], bl0)), l1 = fmaf(u1, wl4[3], fmaf(u0, wl4[1], bl1));
        float yv[CO];
#pragma unroll
        for (int o = 0; o < CO; ++o) yv[o] = fmaf(l1, w_[(CX + CS + 1) * CO + o], fmaf(l0, w_[(CX + CS) * CO + o], sy[o]));
#pragma unroll
        for (int i = 0; i < CX; ++i)
#pragma unroll
            for (int o = 0; o < CO; ++o) yv[o] = fmaf(xv[i], w_[i * CO + o], yv[o]);
#pragma unroll
        for (int o = 0; o < CO; ++o) ys[tid * CO + o] = fmaxf(yv[o], 0.f);
        *reinterpret_cast<float2*>(a.u + (px0 + tid) * 2) = make_float2(u0, u1);
        *reinterpret_cast<float2*>(a.lws + (px0 + tid) * 2) = make_float2(l0, l1);
        __syncthreads();
        f32x4* dst = reinterpret_cast<f32x4*>(a.y + px0 * CO);
#pragma unroll
        for (int k = 0; k < YIT; ++k) {
            const int e = k * 256 + tid;
            if (e < NY4) dst[e] = *reinterpret_cast<const f32x4*>(ys + e * 4);
        }
    }
}

struct TailBwdParams {
    const float *x, *s, *u, *lws, *y, *dy, *wt, *wl, *w;
    float *dx, *ds, *dz, *dzs, *dwl_part, *dbl_part;      // dz: [B T][HW][CO + 2], dzs: [B][HW][CO + 2] (summed over t)
    float* wpart;            // staged form: [blocks][2][16][16] partial 1x1 weight gradients (dz is never stored)
    int B, T, HW, acc_dx, acc_ds, want_dx, want_ds;
    int mask_x, mask_s;      // x / s are ReLU outputs whose backward mask their consumers apply (GTensor::grad_masked): zero dx where x <= 0
};

template <int CX, int CS, int CO>
__global__ void __launch_bounds__(256) rec_tail_bwd_kernel(const TailBwdParams a) {
    cfloat_p wt_ = uniform_ro(a.wt), w_ = uniform_ro(a.w);
    constexpr int CZ = CO + 2;
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= (size_t)a.B * a.HW) return;
    const int bi = (int)(p / a.HW), hw = (int)(p - (size_t)bi * a.HW);
    const f32x4 wl4 = *reinterpret_cast<const f32x4*>(a.wl + (size_t)hw * 4);
    float zsum[CZ], dsacc[CS], dwl[4] = {0.f, 0.f, 0.f, 0.f}, dbl[2] = {0.f, 0.f};
#pragma unroll
    for (int o = 0; o < CZ; ++o) zsum[o] = 0.f;
#pragma unroll
    for (int j = 0; j < CS; ++j) dsacc[j] = 0.f;
    const size_t f0 = (size_t)bi * a.T * a.HW + hw;                  // the sample's first frame at this grid point
    // descriptors based at the block's first sample / first grid point (see rec_tail_fwd_kernel): offsets independent of the batch size
    const size_t p0 = (size_t)blockIdx.x * 256, fbase = (p0 / a.HW) * a.T * a.HW;
    const __amdgpu_buffer_rsrc_t ry = rsrc_of(a.y + fbase * CO), rdy = rsrc_of(a.dy + fbase * CO), rdz = rsrc_of(a.dz + fbase * CZ),
                                 rzs = rsrc_of(a.dzs + p0 * CZ);
    for (int t = 0; t < a.T; ++t) {
        const size_t px = f0 + (size_t)t * a.HW;
        float z[CZ];                                                 // dy masked by y > 0, then the transition's masked gradient
        {
            float dyv[CO], yv[CO];
            load_row<CO>(rdy, (int)((px - fbase) * CO * 4), dyv);
            load_row<CO>(ry, (int)((px - fbase) * CO * 4), yv);
#pragma unroll
            for (int o = 0; o < CO; ++o) z[o] = yv[o] > 0.f ? dyv[o] : 0.f;
        }
        const float2 uv = *reinterpret_cast<const float2*>(a.u + px * 2);
        float dl0 = 0.f, dl1 = 0.f;
#pragma unroll
        for (int o = 0; o < CO; ++o) { dl0 = fmaf(w_[(CX + CS) * CO + o], z[o], dl0); dl1 = fmaf(w_[(CX + CS + 1) * CO + o], z[o], dl1); }
        // lws[f] = sum_c u[c] Wl[c][f] + bl[f]
        dwl[0] = fmaf(uv.x, dl0, dwl[0]); dwl[1] = fmaf(uv.x, dl1, dwl[1]); dwl[2] = fmaf(uv.y, dl0, dwl[2]); dwl[3] = fmaf(uv.y, dl1, dwl[3]);
        dbl[0] += dl0; dbl[1] += dl1;
        const float du0 = fmaf(wl4[1], dl1, wl4[0] * dl0), du1 = fmaf(wl4[3], dl1, wl4[2] * dl0);
        z[CO] = uv.x > 0.f ? du0 : 0.f;
        z[CO + 1] = uv.y > 0.f ? du1 : 0.f;
        if (a.want_dx) {
            float dxv[CX], xm[CX];
            if (a.mask_x) {
#pragma unroll
                for (int q = 0; q < CX; q += 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(a.x + px * CX + q);
                    xm[q] = v[0]; xm[q + 1] = v[1]; xm[q + 2] = v[2]; xm[q + 3] = v[3];
                }
            }
#pragma unroll
            for (int i = 0; i < CX; ++i) {
                float v = fmaf(wt_[i * 2 + 1], z[CO + 1], wt_[i * 2] * z[CO]);
#pragma unroll
                for (int o = 0; o < CO; ++o) v = fmaf(w_[i * CO + o], z[o], v);
                dxv[i] = (a.mask_x && !(xm[i] > 0.f)) ? 0.f : v;
            }
#pragma unroll
            for (int q = 0; q < CX; q += 4) {
                f32x4 v = {dxv[q], dxv[q + 1], dxv[q + 2], dxv[q + 3]};
                f32x4* d = reinterpret_cast<f32x4*>(a.dx + px * CX + q);
                if (a.acc_dx) v += *d;
                *d = v;
            }
        }
#pragma unroll
        for (int j = 0; j < CS; ++j) {
            float v = fmaf(wt_[(CX + j) * 2 + 1], z[CO + 1], wt_[(CX + j) * 2] * z[CO]);
#pragma unroll
            for (int o = 0; o < CO; ++o) v = fmaf(w_[(CX + j) * CO + o], z[o], v);
            dsacc[j] += v;
        }
        store_row<CZ>(rdz, (int)((px - fbase) * CZ * 4), z);
#pragma unroll
        for (int o = 0; o < CZ; ++o) zsum[o] += z[o];
    }
    store_row<CZ>(rzs, (int)((p - p0) * CZ * 4), zsum);
    if (a.want_ds) {
#pragma unroll
        for (int q = 0; q < CS; q += 4) {
            f32x4 v = {dsacc[q], dsacc[q + 1], dsacc[q + 2], dsacc[q + 3]};
            if (a.mask_s) {
                const f32x4 m = *reinterpret_cast<const f32x4*>(a.s + p * CS + q);
                v[0] = m[0] > 0.f ? v[0] : 0.f; v[1] = m[1] > 0.f ? v[1] : 0.f; v[2] = m[2] > 0.f ? v[2] : 0.f; v[3] = m[3] > 0.f ? v[3] : 0.f;
            }
            f32x4* d = reinterpret_cast<f32x4*>(a.ds + p * CS + q);
            if (a.acc_ds) v += *d;
            *d = v;
        }
    }
    *reinterpret_cast<f32x4*>(a.dwl_part + p * 4) = (f32x4){dwl[0], dwl[1], dwl[2], dwl[3]};
    *reinterpret_cast<float2*>(a.dbl_part + p * 2) = make_float2(dbl[0], dbl[1]);
}

// ---- backward, coalesced in the same way: dy, y, x arrive as contiguous streams into LDS, dx leaves as one; u / lws of the
// next frame are in flight with them.  The 1x1 weight gradients are taken HERE, on the matrix pipe, from what already sits in
// LDS: per frame a wave multiplies its 64 pixels' rows  [x (CX) | s (CS), lws (2), 1]^T (two 16-row A tiles)  by  dz (CO + 2
// columns, B tile) with pixels as the k index -- 16 k-steps x 2 v_mfma_f32_16x16x4_f32 -- and keeps the two 16 x 16 sums in
// eight registers over all T frames.  The row of ones yields the bias gradients, s rides along every frame (same sum as
// s x sum_t dz).  dz is never written to HBM and x / dz are not read a second time by separate weight-gradient launches
// (cfg4: three conv_wgrad_rows<1,1,1,1> launches, 0.62 ms per step, gone).  Per block one [2][16][16] partial, summed in a
// fixed order by rec_tail_wsum_kernel and the finishing kernel.
constexpr int REC_TAIL_WTILE = 512;
template <int CX, int CS, int CO>
__global__ void __launch_bounds__(256) rec_tail_bwd_staged_kernel(const TailBwdParams a) {
    cfloat_p wt_ = uniform_ro(a.wt), w_ = uniform_ro(a.w);
    constexpr int CZ = CO + 2, XQ = CX / 4, XP = CX + 4, AP = CS + 4;
    static_assert(CZ <= 16 && CX <= 16 && CS + 3 <= 16 && 4 * REC_TAIL_WTILE <= 256 * CO, "rec_tail: tile shapes");
    constexpr int NR4 = 256 * CO / 4, RIT = (NR4 + 255) / 256;       // float4 per block row-stream of a CO-channel tensor
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const dys = lds;                      // [256][CO]   dy masked by y > 0 (at load time)
    float* const dzs_ = dys + 256 * CO;          // [256][CZ]
    float* const dxs = dzs_ + 256 * CZ;          // [256][XP]   (dx out)
    float* const xss = dxs + 256 * XP;           // [256][XP]   (x in)
    float* const auxs = xss + 256 * XP;          // [256][AP]   s, lws, 1, 0: the second A tile's rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
    const int blocks_per_sample = a.HW / 256;
    const int bi = blockIdx.x / blocks_per_sample, hw0 = (blockIdx.x - bi * blocks_per_sample) * 256, hw = hw0 + tid;
    const size_t p = (size_t)bi * a.HW + hw;
    const f32x4 wl4 = *reinterpret_cast<const f32x4*>(a.wl + (size_t)hw * 4);
    float dsacc[CS], dwl[4] = {0.f, 0.f, 0.f, 0.f}, dbl[2] = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < CS; ++j) dsacc[j] = 0.f;
#pragma unroll
    for (int q = 0; q < CS; q += 4) *reinterpret_cast<f32x4*>(auxs + tid * AP + q) = *reinterpret_cast<const f32x4*>(a.s + p * CS + q);
    auxs[tid * AP + CS + 2] = 1.f;
    auxs[tid * AP + CS + 3] = 0.f;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    f32x4 rin[2][RIT], xin[XQ];
    float2 uvn, lwn;
    auto fetch = [&](int t) __attribute__((always_inline)) {
        const size_t px0 = ((size_t)bi * a.T + t) * a.HW + hw0;
        const f32x4* sdy = reinterpret_cast<const f32x4*>(a.dy + px0 * CO);
        const f32x4* sy = reinterpret_cast<const f32x4*>(a.y + px0 * CO);
#pragma unroll
        for (int k = 0; k < RIT; ++k) {
            const int e = k * 256 + tid;
            if (e < NR4) { rin[0][k] = sdy[e]; rin[1][k] = sy[e]; }
        }
        const f32x4* sx = reinterpret_cast<const f32x4*>(a.x + px0 * CX);
#pragma unroll
        for (int k = 0; k < XQ; ++k) xin[k] = sx[k * 256 + tid];
        uvn = *reinterpret_cast<const float2*>(a.u + (px0 + tid) * 2);
        lwn = *reinterpret_cast<const float2*>(a.lws + (px0 + tid) * 2);
    };
    fetch(0);
    for (int t = 0; t < a.T; ++t) {
        const size_t px0 = ((size_t)bi * a.T + t) * a.HW + hw0;
#pragma unroll
        for (int k = 0; k < RIT; ++k) {
            const int e = k * 256 + tid;
            if (e < NR4) {
                f32x4 d = rin[0][k];
                const f32x4 y = rin[1][k];
                d[0] = y[0] > 0.f ? d[0] : 0.f; d[1] = y[1] > 0.f ? d[1] : 0.f; d[2] = y[2] > 0.f ? d[2] : 0.f; d[3] = y[3] > 0.f ? d[3] : 0.f;
                *reinterpret_cast<f32x4*>(dys + e * 4) = d;
            }
        }
#pragma unroll
        for (int k = 0; k < XQ; ++k) {
            const int e = k * 256 + tid;
            *reinterpret_cast<f32x4*>(xss + (e / XQ) * XP + (e % XQ) * 4) = xin[k];
        }
        const float2 uv = uvn, lw = lwn;
        *reinterpret_cast<float2*>(auxs + tid * AP + CS) = lw;
        __syncthreads();
        if (t + 1 < a.T) fetch(t + 1);
        REC_TAIL_OPAQUE(wt_); REC_TAIL_OPAQUE(w_);
        float z[CZ];
#pragma unroll
        for (int o = 0; o < CO; ++o) z[o] = dys[tid * CO + o];
        float dl0 = 0.f, dl1 = 0.f;
#pragma unroll
        for (int o = 0; o < CO; ++o) { dl0 = fmaf(w_[(CX + CS) * CO + o], z[o], dl0); dl1 = fmaf(w_[(CX + CS + 1) * CO + o], z[o], dl1); }
        dwl[0] = fmaf(uv.x, dl0, dwl[0]); dwl[1] = fmaf(uv.x, dl1, dwl[1]); dwl[2] = fmaf(uv.y, dl0, dwl[2]); dwl[3] = fmaf(uv.y, dl1, dwl[3]);
        dbl[0] += dl0; dbl[1] += dl1;
        const float du0 = fmaf(wl4[1], dl1, wl4[0] * dl0), du1 = fmaf(wl4[3], dl1, wl4[2] * dl0);
        z[CO] = uv.x > 0.f ? du0 : 0.f;
        z[CO + 1] = uv.y > 0.f ? du1 : 0.f;
#pragma unroll
        for (int o = 0; o < CZ; ++o) dzs_[tid * CZ + o] = z[o];
        // the wave's own 64 pixel rows of xss / auxs / dzs_ are complete (written by this wave, or before the barrier): LDS
        // operations of a wave execute in order
        {
            const int cb = l15 < CZ ? l15 : CZ - 1, c1 = l15 < AP ? l15 : AP - 1, c0 = l15 < XP ? l15 : XP - 1;
#pragma unroll 4
            for (int ks = 0; ks < 16; ++ks) {
                const int pix = wave * 64 + 4 * ks + lq;
                float bv = dzs_[pix * CZ + cb], a0 = xss[pix * XP + c0], a1 = auxs[pix * AP + c1];
                bv = l15 < CZ ? bv : 0.f;
                a0 = l15 < CX ? a0 : 0.f;
                a1 = l15 < CS + 3 ? a1 : 0.f;
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv, acc1, 0, 0, 0);
            }
        }
        if (a.want_dx) {
#pragma unroll
            for (int q = 0; q < CX; q += 4) {
                f32x4 d4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = q + r;
                    float v = fmaf(wt_[i * 2 + 1], z[CO + 1], wt_[i * 2] * z[CO]);
#pragma unroll
                    for (int o = 0; o < CO; ++o) v = fmaf(w_[i * CO + o], z[o], v);
                    if (a.mask_x && !(xss[tid * XP + i] > 0.f)) v = 0.f;
                    d4[r] = v;
                }
                *reinterpret_cast<f32x4*>(dxs + tid * XP + q) = d4;
            }
        }
#pragma unroll
        for (int j = 0; j < CS; ++j) {
            float v = fmaf(wt_[(CX + j) * 2 + 1], z[CO + 1], wt_[(CX + j) * 2] * z[CO]);
#pragma unroll
            for (int o = 0; o < CO; ++o) v = fmaf(w_[(CX + j) * CO + o], z[o], v);
            dsacc[j] += v;
        }
        __syncthreads();
        if (a.want_dx) {
            f32x4* dxo = reinterpret_cast<f32x4*>(a.dx + px0 * CX);
#pragma unroll
            for (int k = 0; k < XQ; ++k) {
                const int e = k * 256 + tid;
                f32x4 v = *reinterpret_cast<const f32x4*>(dxs + (e / XQ) * XP + (e % XQ) * 4);
                if (a.acc_dx) v += dxo[e];
                dxo[e] = v;
            }
        }
    }
    if (a.want_ds) {
#pragma unroll
        for (int q = 0; q < CS; q += 4) {
            f32x4 v = {dsacc[q], dsacc[q + 1], dsacc[q + 2], dsacc[q + 3]};
            if (a.mask_s) {
                const f32x4 m = *reinterpret_cast<const f32x4*>(auxs + tid * AP + q);
                v[0] = m[0] > 0.f ? v[0] : 0.f; v[1] = m[1] > 0.f ? v[1] : 0.f; v[2] = m[2] > 0.f ? v[2] : 0.f; v[3] = m[3] > 0.f ? v[3] : 0.f;
            }
            f32x4* d = reinterpret_cast<f32x4*>(a.ds + p * CS + q);
            if (a.acc_ds) v += *d;
            *d = v;
        }
    }
    *reinterpret_cast<f32x4*>(a.dwl_part + p * 4) = (f32x4){dwl[0], dwl[1], dwl[2], dwl[3]};
    *reinterpret_cast<float2*>(a.dbl_part + p * 2) = make_float2(dbl[0], dbl[1]);
    // the four waves' tiles -> one partial per block (lane: column l15, rows 4 lq + r)
    __syncthreads();
    float* const red = dys;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        red[wave * REC_TAIL_WTILE + (4 * lq + r) * 16 + l15] = acc0[r];
        red[wave * REC_TAIL_WTILE + 256 + (4 * lq + r) * 16 + l15] = acc1[r];
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int e = h * 256 + tid;
        a.wpart[(size_t)blockIdx.x * REC_TAIL_WTILE + e] = (red[e] + red[REC_TAIL_WTILE + e]) + (red[2 * REC_TAIL_WTILE + e] + red[3 * REC_TAIL_WTILE + e]);
    }
}

// first stage of the fixed-order sum over the blocks' partial tiles: group g adds partials g, g + G, g + 2G, ...
__global__ void __launch_bounds__(256) rec_tail_wsum_kernel(const float* __restrict__ part, int nblk, float* __restrict__ out) {
    const int g = blockIdx.x, G = gridDim.x, tid = threadIdx.x;
    float v0 = 0.f, v1 = 0.f;
    for (int b = g; b < nblk; b += G) {
        v0 += part[(size_t)b * REC_TAIL_WTILE + tid];
        v1 += part[(size_t)b * REC_TAIL_WTILE + 256 + tid];
    }
    out[(size_t)g * REC_TAIL_WTILE + tid] = v0;
    out[(size_t)g * REC_TAIL_WTILE + 256 + tid] = v1;
}

// the parameter gradients from the pieces: dWl / dbl = sum over the samples of the per-sample partials (fixed order);
// dW rows = [x-part | s-part | lws-part] columns 0 .. CO-1 of the three 1x1 weight gradients, dWt rows = columns CO, CO+1 of the
// first two; db / dbt = the first call's bias gradient
struct TailFinishParams {
    const float *dwl_part, *dbl_part, *gx, *gs, *gl, *gb;           // gx [CX][CZ], gs [CS][CZ], gl [2][CZ], gb [CZ]
    const float* tiles;      // or (staged form): [G][2][16][16] sums of rec_tail_wsum_kernel; tile 0 rows = x, tile 1 rows = s, lws, 1
    float *dwl, *dbl, *dw, *db, *dwt, *dbt;
    int B, HW, CX, CS, CO, acc, G;
};
__global__ void __launch_bounds__(256) rec_tail_finish_kernel(const TailFinishParams a) {
    const int CZ = a.CO + 2;
    const size_t n_l = (size_t)a.HW * 6;
    const size_t n_w = (size_t)(a.CX + a.CS + 2) * a.CO + a.CO + (size_t)(a.CX + a.CS) * 2 + 2;
    auto tile = [&](int tl, int row, int col) {
        // (eight loads in flight, added in the order of g: the same sum as a plain walk)
        const float* q = a.tiles + tl * 256 + row * 16 + col;
        float v = 0.f;
        int g = 0;
        for (; g + 8 <= a.G; g += 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = q[(size_t)(g + u) * REC_TAIL_WTILE];
#pragma unroll
            for (int u = 0; u < 8; ++u) v += t[u];
        }
        for (; g < a.G; ++g) v += q[(size_t)g * REC_TAIL_WTILE];
        return v;
    };
    auto GX = [&](int row, int col) { return a.tiles ? tile(0, row, col) : a.gx[row * CZ + col]; };
    auto GS = [&](int row, int col) { return a.tiles ? tile(1, row, col) : a.gs[row * CZ + col]; };
    auto GL = [&](int row, int col) { return a.tiles ? tile(1, a.CS + row, col) : a.gl[row * CZ + col]; };
    auto GB = [&](int col) { return a.tiles ? tile(1, a.CS + 2, col) : a.gb[col]; };
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n_l + n_w; e += (size_t)gridDim.x * 256) {
        if (e < n_l) {
            const size_t hw = e / 6;
            const int k = (int)(e - hw * 6);
            float v = 0.f;
            for (int b = 0; b < a.B; ++b)
                v += k < 4 ? a.dwl_part[((size_t)b * a.HW + hw) * 4 + k] : a.dbl_part[((size_t)b * a.HW + hw) * 2 + (k - 4)];
            float* d = k < 4 ? a.dwl + hw * 4 + k : a.dbl + hw * 2 + (k - 4);
            *d = a.acc ? *d + v : v;
            continue;
        }
        size_t r = e - n_l;
        float v;
        float* d;
        const size_t nW = (size_t)(a.CX + a.CS + 2) * a.CO;
        if (r < nW) {
            const int row = (int)(r / a.CO), o = (int)(r - (size_t)row * a.CO);
            v = row < a.CX ? GX(row, o) : (row < a.CX + a.CS ? GS(row - a.CX, o) : GL(row - a.CX - a.CS, o));
            d = a.dw + r;
        } else if ((r -= nW) < (size_t)a.CO) {
            v = GB((int)r); d = a.db + r;
        } else if ((r -= a.CO) < (size_t)(a.CX + a.CS) * 2) {
            const int row = (int)(r >> 1), c = (int)(r & 1);
            v = row < a.CX ? GX(row, a.CO + c) : GS(row - a.CX, a.CO + c);
            d = a.dwt + r;
        } else {
            r -= (size_t)(a.CX + a.CS) * 2;
            v = GB(a.CO + (int)r); d = a.dbt + r;
        }
        *d = a.acc ? *d + v : v;
    }
}

bool rec_tail_staged(int HW) {
    static const bool no_staged = exp_env("DL4DS_REC_TAIL_NO_STAGED") != nullptr;
    return HW % 256 == 0 && !no_staged;
}
template <int CX, int CS, int CO>
void launch_fwd(hipStream_t s, const TailParams& p) {
    if (rec_tail_staged(p.HW)) {
        DL4DS_LAUNCH((rec_tail_fwd_staged_kernel<CX, CS, CO>), dim3((unsigned)((size_t)p.B * p.HW / 256)), dim3(256), 0, s, p);
        HIP_CHECK(hipGetLastError());
        return;
    }
    const size_t n = (size_t)p.B * p.HW;
    DL4DS_LAUNCH((rec_tail_fwd_kernel<CX, CS, CO>), dim3((unsigned)cdivz(n, 256)), dim3(256), 0, s, p);
    HIP_CHECK(hipGetLastError());
}
template <int CX, int CS, int CO>
void launch_bwd(hipStream_t s, const TailBwdParams& p) {
    if (rec_tail_staged(p.HW)) {
        const size_t lds = (size_t)256 * (CO + (CO + 2) + 2 * (CX + 4) + (CS + 4)) * sizeof(float);
        static bool attr = false;
        if (!attr) {
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(rec_tail_bwd_staged_kernel<CX, CS, CO>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr = true;
        }
        DL4DS_LAUNCH((rec_tail_bwd_staged_kernel<CX, CS, CO>), dim3((unsigned)((size_t)p.B * p.HW / 256)), dim3(256), lds, s, p);
        HIP_CHECK(hipGetLastError());
        return;
    }
    const size_t n = (size_t)p.B * p.HW;
    DL4DS_LAUNCH((rec_tail_bwd_kernel<CX, CS, CO>), dim3((unsigned)cdivz(n, 256)), dim3(256), 0, s, p);
    HIP_CHECK(hipGetLastError());
}

#define REC_TAIL_SHAPES(X) X(16, 8, 13) X(8, 8, 9) X(16, 8, 8) X(8, 8, 8)

struct RecTailOp : GOp {
    int x, s, out, T;
    int wt, bt, wl, bl, w, b;
    int CX, CS, CO;
    RecTailOp() { kind = "rec_tail"; }
    size_t frames(Graph& g, int B) const { return (size_t)B * T; }
    // The non-staged kernels (grids whose point count is no multiple of 256) address rows with 32-bit byte offsets from a descriptor
    // based at the block's first sample: a block of 256 grid points spans at most 256 / HW + 2 samples.  Independent of the batch
    // size (the round-4 check bounded B T HW and also stopped the staged kernels, which index with size_t).
    void require_row_offsets_fit(int HW) const {
        if (rec_tail_staged(HW)) return;
        const size_t span = (size_t)(256 / HW + 2) * T * HW * (size_t)(CO + 2) * 4;
        DL4DS_REQUIRE(span < (1ull << 31), "rec_tail: one sample's frames exceed 32-bit row offsets (DL4DS_NO_REC_TAIL_FUSION=1 builds the separate layers)");
    }
    size_t saved_floats_per_sample(Graph& g) override { return 4 * (size_t)T * g.tensors[out].H * g.tensors[out].W; }   // u, lws
    // backward: dz, dzs, the per-sample LocallyConnected partials, the three small weight-gradient results, then the 1x1
    // weight-gradient kernels' own workspace
    struct Carve { float *dz, *dzs, *pl, *pb, *gx, *gs, *gl, *gb, *wpart, *wred, *rest; size_t rest_bytes; };
    static constexpr int WSUM_GROUPS = 64;
    size_t carve_floats(Graph& g, int B) const {
        const size_t HW = (size_t)g.tensors[out].H * g.tensors[out].W, CZ = CO + 2;
        auto up = [](size_t v) { return (v + 3) & ~(size_t)3; };
        if (rec_tail_staged((int)HW))           // no dz / dzs: the weight gradients come out of the backward kernel as tiles
            return up((size_t)B * HW * 4) + up((size_t)B * HW * 2) + (size_t)B * (HW / 256) * REC_TAIL_WTILE + (size_t)WSUM_GROUPS * REC_TAIL_WTILE;
        return up(frames(g, B) * HW * CZ) + up((size_t)B * HW * CZ) + up((size_t)B * HW * 4) + up((size_t)B * HW * 2) + up((size_t)CX * CZ) +
               up((size_t)CS * CZ) + up(2 * CZ) + up(CZ);
    }
    Carve carve(Graph& g, int B, float* ws, size_t ws_bytes) const {
        const size_t HW = (size_t)g.tensors[out].H * g.tensors[out].W, CZ = CO + 2;
        auto up = [](size_t v) { return (v + 3) & ~(size_t)3; };
        Carve c = {};
        float* p = ws;
        if (rec_tail_staged((int)HW)) {
            c.pl = p; p += up((size_t)B * HW * 4);
            c.pb = p; p += up((size_t)B * HW * 2);
            c.wpart = p; p += (size_t)B * (HW / 256) * REC_TAIL_WTILE;
            c.wred = p; p += (size_t)WSUM_GROUPS * REC_TAIL_WTILE;
            c.rest = p;
            DL4DS_REQUIRE((size_t)(p - ws) * sizeof(float) <= ws_bytes, "rec_tail: workspace too small");
            c.rest_bytes = ws_bytes - (size_t)(p - ws) * sizeof(float);
            return c;
        }
        c.dz = p; p += up(frames(g, B) * HW * CZ);
        c.dzs = p; p += up((size_t)B * HW * CZ);
        c.pl = p; p += up((size_t)B * HW * 4);
        c.pb = p; p += up((size_t)B * HW * 2);
        c.gx = p; p += up((size_t)CX * CZ);
        c.gs = p; p += up((size_t)CS * CZ);
        c.gl = p; p += up(2 * CZ);
        c.gb = p; p += up(CZ);
        c.rest = p;
        DL4DS_REQUIRE((size_t)(p - ws) * sizeof(float) <= ws_bytes, "rec_tail: workspace too small");
        c.rest_bytes = ws_bytes - (size_t)(p - ws) * sizeof(float);
        return c;
    }
    size_t workspace_bytes(Graph& g, int B) override {
        const GTensor& to = g.tensors[out];
        const int N = (int)frames(g, B), CZ = CO + 2;
        TView xv = make_view(nullptr, N, to.H, to.W, CX), zv = make_view(nullptr, N, to.H, to.W, CZ);
        TView sv = make_view(nullptr, B, to.H, to.W, CS), zs = make_view(nullptr, B, to.H, to.W, CZ);
        TView lv = make_view(nullptr, N, to.H, to.W, 2);
        const size_t wg = std::max(std::max(conv2d_wgrad_workspace_bytes(xv, zv, 1), conv2d_wgrad_workspace_bytes(sv, zs, 1)),
                                   conv2d_wgrad_workspace_bytes(lv, zv, 1));
        return carve_floats(g, B) * sizeof(float) + wg + 256;
    }
    void forward(Graph& g, int B, bool) override {
        const GTensor& to = g.tensors[out];
        TailParams p;
        p.x = g.tensors[x].data; p.s = g.tensors[s].data;
        p.wt = g.wp(wt); p.bt = g.wp(bt); p.wl = g.wp(wl); p.bl = g.wp(bl); p.w = g.wp(w); p.b = g.wp(b);
        p.y = to.data;
        p.u = saved; p.lws = saved + 2 * frames(g, B) * to.H * to.W;
        p.B = B; p.T = T; p.HW = to.H * to.W;
        require_row_offsets_fit(p.HW);
        const double px = (double)frames(g, B) * p.HW;
        ProfScope ps(g.stream, "rec_tail_fwd", 2.0 * px * ((CX + CS) * 2 + 4 + (CX + CS + 2) * CO), 4.0 * px * (CX + CO + 4) + 4.0 * B * p.HW * CS);
#define X(A_, B_, C_) if (CX == A_ && CS == B_ && CO == C_) { launch_fwd<A_, B_, C_>(g.stream, p); return; }
        REC_TAIL_SHAPES(X)
#undef X
        DL4DS_REQUIRE(false, "rec_tail: shape not built");
    }
    void backward(Graph& g, const BwdCtx& c) override {
        if (!g.tensors[out].grad_written) return;
        DL4DS_REQUIRE(c.param_grads, "rec_tail: backward without parameter gradients is not supported");
        const GTensor& to = g.tensors[out];
        const int cnt = c.b_cnt < 0 ? c.B : c.b_cnt, HW = to.H * to.W, CZ = CO + 2;
        const size_t f_off = (size_t)c.b_off * T * HW, fr = (size_t)cnt * T;
        Carve cv = carve(g, cnt, g.workspace, g.workspace_bytes);
        TailBwdParams p;
        p.x = g.tensors[x].data + f_off * CX;
        p.s = g.tensors[s].data + (size_t)c.b_off * HW * CS;
        p.mask_x = g.tensors[x].grad_masked; p.mask_s = g.tensors[s].grad_masked;
        p.u = saved + f_off * 2;
        const float* lws = saved + 2 * frames(g, c.B) * HW + f_off * 2;
        p.lws = lws;
        const bool staged = rec_tail_staged(HW);
        require_row_offsets_fit(HW);
        p.y = to.data + f_off * CO; p.dy = to.grad + f_off * CO;
        p.wt = g.wp(wt); p.wl = g.wp(wl); p.w = g.wp(w);
        p.want_dx = wants_grad(g, x, c); p.want_ds = wants_grad(g, s, c);
        p.dx = p.want_dx ? g.tensors[x].grad + f_off * CX : nullptr;
        p.ds = p.want_ds ? g.tensors[s].grad + (size_t)c.b_off * HW * CS : nullptr;
        p.acc_dx = g.tensors[x].grad_written; p.acc_ds = g.tensors[s].grad_written;
        p.dz = cv.dz; p.dzs = cv.dzs; p.dwl_part = cv.pl; p.dbl_part = cv.pb; p.wpart = cv.wpart;
        p.B = cnt; p.T = T; p.HW = HW;
        {
            const double px = (double)fr * HW;
            // staged: + the weight-gradient products, no dz; x always read
            ProfScope ps(g.stream, "rec_tail_bwd", 2.0 * px * (2 * CO + 8 + (CX + CS) * (CO + 2) + (staged ? (CX + CS + 3) * CZ : 0)),
                         staged ? 4.0 * px * (CX * (p.want_dx ? 2 : 1) + 2 * CO + 4) + 4.0 * cnt * HW * (2 * CS + 6)
                                : 4.0 * px * (CX * (p.want_dx ? 2 : 1) + 2 * CO + CZ + 2) + 4.0 * cnt * HW * (CZ + CS + 6));
#define X(A_, B_, C_) if (CX == A_ && CS == B_ && CO == C_) launch_bwd<A_, B_, C_>(g.stream, p);
            REC_TAIL_SHAPES(X)
#undef X
        }
        if (p.want_dx) g.tensors[x].grad_written = true;
        if (p.want_ds) g.tensors[s].grad_written = true;
        TailFinishParams f = {};
        if (staged) {
            const int nblk = cnt * (HW / 256), G = std::min(nblk, (int)WSUM_GROUPS);
            ProfScope ps(g.stream, "rec_tail_wsum", 0.0, 4.0 * REC_TAIL_WTILE * (nblk + G));
            DL4DS_LAUNCH(rec_tail_wsum_kernel, dim3(G), dim3(256), 0, g.stream, cv.wpart, nblk, cv.wred);
            HIP_CHECK(hipGetLastError());
            f.tiles = cv.wred; f.G = G;
        } else {
        // the three 1x1 weight gradients against dz (bias gradient = column sums of dz: rides on the first)
        TView xv = make_view(const_cast<float*>(p.x), (int)fr, to.H, to.W, CX), zv = make_view(cv.dz, (int)fr, to.H, to.W, CZ);
        TView sv = make_view(g.tensors[s].data + (size_t)c.b_off * HW * CS, cnt, to.H, to.W, CS), zs = make_view(cv.dzs, cnt, to.H, to.W, CZ);
        TView lv = make_view(const_cast<float*>(lws), (int)fr, to.H, to.W, 2);
        conv2d_wgrad(g.stream, xv, zv, 1, cv.gx, 0, cv.gb, 0, cv.rest, cv.rest_bytes);
        conv2d_wgrad(g.stream, sv, zs, 1, cv.gs, 0, nullptr, 0, cv.rest, cv.rest_bytes);
        conv2d_wgrad(g.stream, lv, zv, 1, cv.gl, 0, nullptr, 0, cv.rest, cv.rest_bytes);
        }
        f.dwl_part = cv.pl; f.dbl_part = cv.pb; f.gx = cv.gx; f.gs = cv.gs; f.gl = cv.gl; f.gb = cv.gb;
        f.dwl = g.gp(wl); f.dbl = g.gp(bl); f.dw = g.gp(w); f.db = g.gp(b); f.dwt = g.gp(wt); f.dbt = g.gp(bt);
        f.B = cnt; f.HW = HW; f.CX = CX; f.CS = CS; f.CO = CO;
        f.acc = g.params[w].grad_written;
        {
            ProfScope ps(g.stream, "rec_tail_finish", 0.0, 4.0 * HW * 6.0 * (cnt + 1));
            DL4DS_LAUNCH(rec_tail_finish_kernel, dim3((unsigned)std::min<size_t>(cdivz((size_t)HW * 6 + 1024, 256), 2048)), dim3(256), 0, g.stream, f);
            HIP_CHECK(hipGetLastError());
        }
        for (int pid : {wt, bt, wl, bl, w, b}) g.params[pid].grad_written = true;
    }
    bool reads_tensor(int t) const override { return t == x || t == s; }
};

}  // namespace

bool rec_tail_supported(int CX, int CS, int CO) {
    if (getenv("DL4DS_NO_REC_TAIL_FUSION")) return false;
#define X(A_, B_, C_) if (CX == A_ && CS == B_ && CO == C_) return true;
    REC_TAIL_SHAPES(X)
#undef X
    return false;
}

int g_rec_tail(Graph& g, int x, int s, int wt, int bt, int wl, int bl, int w, int b, int T, int CO) {
    const GTensor tx = g.tensors.at(x), ts = g.tensors.at(s);
    DL4DS_REQUIRE(tx.nmul == T && ts.nmul == 1 && tx.H == ts.H && tx.W == ts.W, "rec_tail: x must be (B,T,H,W,CX), s (B,H,W,CS) on the same grid");
    DL4DS_REQUIRE(rec_tail_supported(tx.C, ts.C, CO), "rec_tail: channel combination not built (see rec_tail_supported)");
    const size_t HW = (size_t)tx.H * tx.W;
    DL4DS_REQUIRE(g.params.at(wt).n == (size_t)(tx.C + ts.C) * 2 && g.params.at(bt).n == 2, "rec_tail: LocalizedConvBlock transition size mismatch");
    DL4DS_REQUIRE(g.params.at(wl).n == HW * 4 && g.params.at(bl).n == HW * 2, "rec_tail: LocallyConnected2D size mismatch");
    DL4DS_REQUIRE(g.params.at(w).n == (size_t)(tx.C + ts.C + 2) * CO && g.params.at(b).n == (size_t)CO, "rec_tail: TransitionLast size mismatch");
    const int out = g.add_tensor(tx.H, tx.W, CO, T, true, false);
    RecTailOp* op = new RecTailOp();
    g.ops.emplace_back(op);
    op->x = x; op->s = s; op->out = out; op->T = T;
    op->wt = wt; op->bt = bt; op->wl = wl; op->bl = bl; op->w = w; op->b = b;
    op->CX = tx.C; op->CS = ts.C; op->CO = CO;
    op->pids = {wt, bt, wl, bl, w, b};
    // (like a Concatenate, this op applies its inputs' ReLU masks to the gradients it writes: n_masking, not n_other)
    g.tensors[x].n_masking++;
    g.tensors[s].n_masking++;
    g.tensors[out].relu_out = true;
    op->out_tid = out; op->in_tids = {x, s};
    g.tensors[out].dep_grad_input = g.tensors[x].dep_grad_input || g.tensors[s].dep_grad_input;
    return out;
}
