// dl4ds_amd -- gfx950 (MI355X / CDNA4) native kernels for the dl4ds conv-SR train step.
// Common device/host helpers.  fp32 NHWC everywhere (Keras channels_last).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <stdexcept>
#include <cstdlib>

// ---- run-time switches (README "Environment switches") ------------------------------------------------------------------------
// The PRODUCT reads eight variables by name through getenv: DL4DS_NO_WINOGRAD, DL4DS_NO_WINOGRAD_WGRAD, DL4DS_NO_REC_TAIL_FUSION,
// DL4DS_NO_CONVLSTM_SEQ (kernel families / graph transformations with an A/B test and a documented fall-back), DL4DS_AUX_STREAM,
// DL4DS_ALLOW_UNSYNCED, DL4DS_COLLECTIVE_TIMEOUT_S, DL4DS_SEQ_RESERVE_CUS (multi-process behaviour).
// test_env: hooks of the test-suite -- force a kernel onto grids it would not pick, play a collective, switch one graph transformation
//   off for an A/B comparison -- honoured only while DL4DS_TEST_HOOKS=1 (tests/conftest.py sets it; a production process never does).
// exp_env: switches of measured-and-dropped variants and of diagnostics.  They exist in -DDL4DS_EXPERIMENTS builds only
//   (DL4DS_BUILD_EXPERIMENTS=1 python dl4ds_amd/csrc/build.py -> dl4ds_amd/libdl4ds_hip_exp.so, loaded through DL4DS_HIP_LIB); in the
//   product library the call is the constant nullptr and the variant's branch is dead code.
#ifdef DL4DS_EXPERIMENTS
inline const char* exp_env(const char* name) { return std::getenv(name); }
#else
inline const char* exp_env(const char*) { return nullptr; }
#endif
inline const char* test_env(const char* name) {
    static const bool hooks = [] { const char* e = std::getenv("DL4DS_TEST_HOOKS"); return e && e[0] == '1'; }();
#ifdef DL4DS_EXPERIMENTS
    (void)hooks;
    return std::getenv(name);
#else
    return hooks ? std::getenv(name) : nullptr;
#endif
}


typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// Tensor view: logical NHWC tensor (N,H,W,C) living in HBM.
//   ld   : floats between consecutive pixels (>= C; > C for channel slices of a wider buffer)
//   d2s  : 0/1 -> plain.  r>=2 -> the memory holds depth_to_space(r) of the logical tensor, i.e.
//          logical (n,h,w,c) lives at [n, h*r+i, w*r+j, c % Cp] of an (N, H*r, W*r, Cp=C/r^2)
//          buffer with (i*r+j) = c / Cp   (tf.nn.depth_to_space "DCR", blocks.py:427).
//   vec  : float4 access at channel offsets that are multiples of 4 is legal (alignment + no
//          group straddling)
struct TView {
    float* p;
    int N, H, W, C;
    int ld;
    int d2s;
    int vec;
    int cp;              // d2s: channels per (i,j) group = C / r^2
    unsigned mcp, mr;    // d2s: magic multipliers for exact n/cp and n/r (fast_div)
    size_t nstride;      // floats between consecutive images n (default: contiguous); lets a view pick
                         // frame t of every sample of a (B,T,H,W,C) buffer without a copy
    // Optional per-image channel affine of a READ view: logical value = mem * sc[n*C + c] + sh[n*C + c] inside the tensor
    // (the SAME zero padding outside it).  ChannelAttention2D's broadcast scale (blocks.py:585-593) and its backward
    // dX = dY * scale + dmean are carried this way into the staging loads of the neighbouring convolutions instead of
    // being materialised by a pass over the HR tensor.  sc == nullptr: plain view; sh may be null on its own (scale only).
    // Only the kernels that say so honour it (conv_direct*, conv_narrow_pair, conv_narrow_wgrad's dz operand); every other
    // dispatch path rejects a view that carries one.
    const float* sc;
    const float* sh;
};

// exact n / d for 0 <= n < 2^20, 1 <= d <= 4096 with magic = 2^32/d + 1 (0 encodes d == 1)
__host__ __device__ inline unsigned div_magic(int d) { return d <= 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)d + 1ull); }
__device__ __forceinline__ int fast_div(int n, unsigned magic) { return magic ? (int)__umulhi((unsigned)n, magic) : n; }

__host__ __device__ inline TView make_view(float* p, int N, int H, int W, int C) {
    TView v;
    v.p = p; v.N = N; v.H = H; v.W = W; v.C = C; v.ld = C; v.d2s = 0;
    v.vec = ((C & 3) == 0) && ((((uintptr_t)p) & 15) == 0);
    v.cp = C; v.mcp = 0; v.mr = 0;
    v.nstride = (size_t)H * W * C;
    v.sc = nullptr; v.sh = nullptr;
    return v;
}

// view whose memory is the depth_to_space(r) image of the logical (N,H,W,C) tensor
__host__ __device__ inline TView make_view_d2s(float* p, int N, int H, int W, int C, int r) {
    TView v;
    int cp = C / (r * r);
    v.p = p; v.N = N; v.H = H; v.W = W; v.C = C; v.ld = cp; v.d2s = r;
    v.vec = ((cp & 3) == 0) && ((((uintptr_t)p) & 15) == 0);
    v.cp = cp; v.mcp = div_magic(cp); v.mr = div_magic(r);
    v.nstride = (size_t)H * W * C;
    v.sc = nullptr; v.sh = nullptr;
    return v;
}

// float4 of the view's channel affine for image n, channels [c, c+4) (c % 4 == 0, C % 4 == 0); identity when absent
__device__ __forceinline__ void view_affine4(const TView& v, int n, int c, float4& s4, float4& h4) {
    s4 = make_float4(1.f, 1.f, 1.f, 1.f);
    h4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (v.sc) {
        const int cs = (c + 3 < v.C) ? c : 0;
        s4 = *reinterpret_cast<const float4*>(v.sc + (size_t)n * v.C + cs);
        if (v.sh) h4 = *reinterpret_cast<const float4*>(v.sh + (size_t)n * v.C + cs);
    }
}
__device__ __forceinline__ float4 affine4(float4 r, const float4& s4, const float4& h4) {
    return make_float4(fmaf(r.x, s4.x, h4.x), fmaf(r.y, s4.y, h4.y), fmaf(r.z, s4.z, h4.z), fmaf(r.w, s4.w, h4.w));
}

__device__ __forceinline__ size_t view_off(const TView& v, int n, int y, int x, int c) {
    if (v.d2s > 1) {
        const int r = v.d2s;
        const int g = fast_div(c, v.mcp);
        const int cc = c - g * v.cp;
        const int i = fast_div(g, v.mr), j = g - i * r;
        return (size_t)n * v.nstride + ((size_t)(y * r + i) * (size_t)(v.W * r) + (x * r + j)) * v.ld + cc;
    }
    return (size_t)n * v.nstride + ((size_t)y * v.W + x) * v.ld + c;
}

// separable addressing: view_off(v,n,y,x,c) == view_pix_base(v,n,y,x) + view_chan_off(v,c)
__device__ __forceinline__ size_t view_pix_base(const TView& v, int n, int y, int x) {
    if (v.d2s > 1) {
        const int r = v.d2s;
        return (size_t)n * v.nstride + (((size_t)y * r) * (size_t)(v.W * r) + (size_t)x * r) * v.ld;
    }
    return (size_t)n * v.nstride + ((size_t)y * v.W + x) * v.ld;
}
__device__ __forceinline__ size_t view_chan_off(const TView& v, int c) {
    if (v.d2s > 1) {
        const int r = v.d2s;
        const int g = fast_div(c, v.mcp);
        const int cc = c - g * v.cp;
        const int i = fast_div(g, v.mr), j = g - i * r;
        return ((size_t)i * (v.W * r) + j) * v.ld + cc;
    }
    return (size_t)c;
}

// channels c..c+3 (c % 4 == 0), zero beyond C
__device__ __forceinline__ float4 view_load4(const TView& v, int n, int y, int x, int c) {
    if (v.vec && c + 3 < v.C) {
        return *reinterpret_cast<const float4*>(v.p + view_off(v, n, y, x, c));
    }
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < v.C) r.x = v.p[view_off(v, n, y, x, c)];
    if (c + 1 < v.C) r.y = v.p[view_off(v, n, y, x, c + 1)];
    if (c + 2 < v.C) r.z = v.p[view_off(v, n, y, x, c + 2)];
    if (c + 3 < v.C) r.w = v.p[view_off(v, n, y, x, c + 3)];
    return r;
}

// Branch-free staging loads.  A load inside a divergent `if` ends its basic block with s_waitcnt vmcnt(0), which
// serialises every staging load into its own L2/HBM round trip (measured: ~40 % of the conv kernels), and a plain
// `ok ? load : 0` select is turned back into exactly that branch by LLVM.  So the load is split in three:
//   *_raw   : ALWAYS issues the load, from a clamped (valid) address, returns whatever is there
//   *_valid : 4-bit mask of the components that are really inside the tensor
//   mask4   : applied later (when the value is written to LDS), through opaque asm so it stays straight-line code
__device__ __forceinline__ float4 mask4(float4 r, unsigned bits) {
    unsigned m0 = (bits & 1u) ? 0xffffffffu : 0u, m1 = (bits & 2u) ? 0xffffffffu : 0u;
    unsigned m2 = (bits & 4u) ? 0xffffffffu : 0u, m3 = (bits & 8u) ? 0xffffffffu : 0u;
    asm volatile("" : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3));
    r.x = __uint_as_float(__float_as_uint(r.x) & m0);
    r.y = __uint_as_float(__float_as_uint(r.y) & m1);
    r.z = __uint_as_float(__float_as_uint(r.z) & m2);
    r.w = __uint_as_float(__float_as_uint(r.w) & m3);
    return r;
}
__device__ __forceinline__ unsigned valid4(int c, int C, bool pred) {
    const int k = pred ? min(max(C - c, 0), 4) : 0;
    return (1u << k) - 1u;
}
__device__ __forceinline__ float4 view_load4_raw(const TView& v, int n, int y, int x, int c, bool pred) {
    const int ys = pred ? y : 0, xs = pred ? x : 0;
    if (v.vec) {                                   // uniform; vec implies C % 4 == 0 (cp % 4 == 0 for d2s)
        const int cs = (pred && c < v.C) ? c : 0;
        return *reinterpret_cast<const float4*>(v.p + view_off(v, n, ys, xs, cs));
    }
    float4 r;
    r.x = v.p[view_off(v, n, ys, xs, min(max(c, 0), v.C - 1))];
    r.y = v.p[view_off(v, n, ys, xs, min(c + 1, v.C - 1))];
    r.z = v.p[view_off(v, n, ys, xs, min(c + 2, v.C - 1))];
    r.w = v.p[view_off(v, n, ys, xs, min(c + 3, v.C - 1))];
    return r;
}
// same for views known to be float4-loadable (v.vec checked by the caller): no scalar arm, a quarter of the registers
__device__ __forceinline__ float4 view_load4_vec(const TView& v, int n, int y, int x, int c, bool pred) {
    const int ys = pred ? y : 0, xs = pred ? x : 0;
    const int cs = (pred && c < v.C) ? c : 0;
    return *reinterpret_cast<const float4*>(v.p + view_off(v, n, ys, xs, cs));
}
// floats [co, co+4) of a row of `Cout` floats
__device__ __forceinline__ float4 row_load4_raw(const float* row, int co, int Cout, bool vec, bool pred) {
    if (vec) return *reinterpret_cast<const float4*>(row + ((pred && co < Cout) ? co : 0));
    float4 r;
    r.x = row[min(max(co, 0), Cout - 1)];
    r.y = row[min(co + 1, Cout - 1)];
    r.z = row[min(co + 2, Cout - 1)];
    r.w = row[min(co + 3, Cout - 1)];
    return r;
}

// ---------------------------------------------------------------------------------------------
// wave / block reductions (wave = 64 lanes on CDNA)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_down(v, o, 64));
    return v;
}

// ---------------------------------------------------------------------------------------------
// host-side error handling: C++ exceptions inside the library, converted to int status + message
// at the C ABI (capi.cpp).
struct Dl4dsError : public std::runtime_error {
    explicit Dl4dsError(const std::string& m) : std::runtime_error(m) {}
};

#define HIP_CHECK(expr)                                                                        \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            throw Dl4dsError(std::string("HIP error: ") + hipGetErrorString(_e) + " at " +     \
                             __FILE__ + ":" + std::to_string(__LINE__) + " (" #expr ")");      \
        }                                                                                      \
    } while (0)

#define DL4DS_REQUIRE(cond, msg)                                                               \
    do {                                                                                       \
        if (!(cond)) throw Dl4dsError(std::string("dl4ds: ") + (msg) + " [" #cond "]");        \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t cdivz(size_t a, size_t b) { return (a + b - 1) / b; }
