// dl4ds_amd -- Winograd F(2x2, 3x3) convolution: eligibility, channel passes, epilogue form (kernel: conv_wino_kernel.h)
#include "conv_wino_kernel.h"
#include "conv_wino4_kernel.h"

namespace {

int wino_cu_count() {
    static const int n = [] {
        int dev = 0, v = 0;
        HIP_CHECK(hipGetDevice(&dev));
        HIP_CHECK(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
        return v;
    }();
    return n;
}

// U = G g G^T in the fragment order the workgroups load it in: element ((pc * 4 + xi) * F/4 + f4) * 64 + lane, component j,
// f = 4 f4 + j = (nu * 4 KQ + ks) * NT + cb (pc = pass * nchunk + chunk, F = 16 KQ NT); nu = 3 negated
// (both forms of the kernel -- conv_wino_kernel / conv_wino2_kernel -- read this one layout: nu = 1, 2 hold (U0 +- U1 + U2) / 2)
__global__ void __launch_bounds__(256) wino_filter_kernel(const float* __restrict__ w, float* __restrict__ u, int Cin, int Cout, int KQ,
                                                          int NT, int nchunk, int total) {
    const int F = 16 * KQ * NT;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int j = idx & 3, lane = (idx >> 2) & 63;
        int r = idx >> 8;
        const int f4 = r % (F / 4); r /= F / 4;
        const int xi = r & 3; r >>= 2;
        const int chunk = r % nchunk, pass = r / nchunk;
        const int f = 4 * f4 + j;
        const int nu = f / (4 * KQ * NT), ks = (f / NT) % (4 * KQ), cb = f % NT;
        const int cin = pass * 16 * KQ + 16 * (ks >> 2) + 4 * (lane >> 4) + (ks & 3);
        const int co = chunk * 16 * NT + 16 * cb + (lane & 15);
        float val = 0.f;
        if (cin < Cin && co < Cout) {
            const float c0 = xi == 0 ? 1.f : (xi == 3 ? 0.f : .5f);
            const float c1 = xi == 1 ? .5f : (xi == 2 ? -.5f : 0.f);
            const float c2 = xi == 3 ? 1.f : (xi == 0 ? 0.f : .5f);
            const size_t tap = (size_t)Cin * Cout;
            const float* p = w + (size_t)cin * Cout + co;
            float t[3];
#pragma unroll
            for (int b = 0; b < 3; ++b) t[b] = c0 * p[(0 * 3 + b) * tap] + c1 * p[(1 * 3 + b) * tap] + c2 * p[(2 * 3 + b) * tap];
            val = nu == 0 ? t[0] : (nu == 1 ? .5f * (t[0] + t[1] + t[2]) : (nu == 2 ? .5f * (t[0] - t[1] + t[2]) : -t[2]));
        }
        u[idx] = val;
    }
}

// F(4x4, 3x3): U = G g G^T (6 x 6 positions, points 0, +-1, +-2, inf) in the order conv_wino4_kernel's waves load it in: element
// ((pc * 4 + wave) * F/4 + f4) * 64 + lane, component j, f = 4 f4 + j = ((x * 3 + n) * 4 KQ + ks) * NT + cb, F = 36 KQ NT; wave (a, b)
// owns positions (xi(a, x), nu(b, n)), xi(0, .) = 0, 1, 2 and xi(1, .) = 5, 3, 4 (the order its transform recipe produces them in), and
// the entry of block 0 that the recipe delivers negated (index 2) carries the sign here.
__device__ __forceinline__ float wino4_filter_value(const float* __restrict__ w, int Cin, int Cout, int KQ, int NT, int nchunk, int idx) {
    const int F = 36 * KQ * NT;
    const int j = idx & 3, lane = (idx >> 2) & 63;
    int r = idx >> 8;
    const int f4 = r % (F / 4); r /= F / 4;
    const int wave = r & 3; r >>= 2;
    const int chunk = r % nchunk, pass = r / nchunk;
    const int f = 4 * f4 + j;
    const int x = f / (12 * KQ * NT), n = (f / (4 * KQ * NT)) % 3, ks = (f / NT) % (4 * KQ), cb = f % NT;
    const int a = wave >> 1, b = wave & 1;
    const int xi = a == 0 ? x : (x == 0 ? 5 : x + 2), nu = b == 0 ? n : (n == 0 ? 5 : n + 2);
    const float sign = ((a == 0 && x == 2) != (b == 0 && n == 2)) ? -1.f : 1.f;
    const int cin = pass * 16 * KQ + 16 * (ks >> 2) + 4 * (lane >> 4) + (ks & 3);
    const int co = chunk * 16 * NT + 16 * cb + (lane & 15);
    if (cin >= Cin || co >= Cout) return 0.f;
    auto G = [](int row, int k) -> float {
        switch (row) {
            case 0: return k == 0 ? .25f : 0.f;
            case 1: return -1.f / 6.f;
            case 2: return k == 1 ? 1.f / 6.f : -1.f / 6.f;
            case 3: return k == 0 ? 1.f / 24.f : (k == 1 ? 1.f / 12.f : 1.f / 6.f);
            case 4: return k == 0 ? 1.f / 24.f : (k == 1 ? -1.f / 12.f : 1.f / 6.f);
            default: return k == 2 ? 1.f : 0.f;
        }
    };
    const size_t tap = (size_t)Cin * Cout;
    const float* p = w + (size_t)cin * Cout + co;
    float t[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) t[q] = G(xi, 0) * p[(0 * 3 + q) * tap] + G(xi, 1) * p[(1 * 3 + q) * tap] + G(xi, 2) * p[(2 * 3 + q) * tap];
    return sign * (G(nu, 0) * t[0] + G(nu, 1) * t[1] + G(nu, 2) * t[2]);
}

__global__ void __launch_bounds__(256) wino4_filter_kernel(const float* __restrict__ w, float* __restrict__ u, int Cin, int Cout, int KQ,
                                                           int NT, int nchunk, int total) {
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x)
        u[idx] = wino4_filter_value(w, Cin, Cout, KQ, NT, nchunk, idx);
}

// transformed filters: grow-only, one buffer per stream (launches on a stream are ordered)
struct WinoScratch { hipStream_t stream; float* buf; size_t floats; };
float* wino_scratch(hipStream_t s, size_t floats) {
    static std::mutex mu;
    static std::vector<WinoScratch> all;
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
    WinoScratch e{s, nullptr, std::max<size_t>(floats, 1 << 20)};
    HIP_CHECK(hipMalloc((void**)&e.buf, e.floats * sizeof(float)));
    all.push_back(e);
    return e.buf;
}

// ---- transformed filters of a GRAPH's layers: one batched launch per pass (round 5) ---------------------------------------------
// Inside a graph pass (WinoPassGuard, opened by Graph::forward / Graph::backward) every Winograd layer registers its filter
// (pointer into the graph's parameter arena W or its derived-weights arena Wt, shape, fragment geometry) with a buffer of its own.
// From the second pass on, wino_filters_refresh(range) transforms ALL registered filters of that range in ONE launch -- at the start
// of the forward pass for W, right after the dgrad arrangements have been rebuilt for Wt -- and the layers find their entry fresh:
// 20 launches of wino_filter_kernel per cfg2 step become 2.  Freshness never outlives a forward pass: Graph::forward invalidates
// the graph's ranges first (the optimiser, set_weights, a checkpoint load or a broadcast may have touched W).  Outside a graph pass
// (the op-level API) nothing is registered or trusted: the shared scratch and one launch per call, as before.
struct WinoFilterJob { const float* w; float* u; int Cin, Cout, KQ, NT, nchunk, total, first, f44; };     // first: the job's first block; f44: F(4x4) order
constexpr int WINO_JOBS_MAX = 24, WINO_FILTER_PER_BLOCK = 1024;
struct WinoFilterJobs { WinoFilterJob j[WINO_JOBS_MAX]; int n, total; };                            // total: blocks

__global__ void __launch_bounds__(256) wino_filter_batched_kernel(const WinoFilterJobs jobs) {
    // block -> job: wave-uniform (scalar loads from the kernel arguments; a per-thread job index made the compiler copy the
    // whole table into scratch memory per thread)
    int k = 0;
#pragma unroll 1
    while (k + 1 < jobs.n && (int)blockIdx.x >= jobs.j[k + 1].first) ++k;
    const float* const w = jobs.j[k].w;
    float* const u = jobs.j[k].u;
    const int Cin = jobs.j[k].Cin, Cout = jobs.j[k].Cout, KQ = jobs.j[k].KQ, NT = jobs.j[k].NT, nchunk = jobs.j[k].nchunk;
    const int total = jobs.j[k].total, b0 = jobs.j[k].first;
    const int F = 16 * KQ * NT;
    if (jobs.j[k].f44) {
#pragma unroll 1
        for (int q = 0; q < WINO_FILTER_PER_BLOCK / 256; ++q) {
            const int idx = ((int)blockIdx.x - b0) * WINO_FILTER_PER_BLOCK + q * 256 + (int)threadIdx.x;
            if (idx >= total) break;
            u[idx] = wino4_filter_value(w, Cin, Cout, KQ, NT, nchunk, idx);
        }
        return;
    }
#pragma unroll 1
    for (int q = 0; q < WINO_FILTER_PER_BLOCK / 256; ++q) {
        const int idx = ((int)blockIdx.x - b0) * WINO_FILTER_PER_BLOCK + q * 256 + (int)threadIdx.x;
        if (idx >= total) break;
        const int j = idx & 3, lane = (idx >> 2) & 63;
        int r = idx >> 8;
        const int f4 = r % (F / 4); r /= F / 4;
        const int xi = r & 3; r >>= 2;
        const int chunk = r % nchunk, pass = r / nchunk;
        const int f = 4 * f4 + j;
        const int nu = f / (4 * KQ * NT), ks = (f / NT) % (4 * KQ), cb = f % NT;
        const int cin = pass * 16 * KQ + 16 * (ks >> 2) + 4 * (lane >> 4) + (ks & 3);
        const int co = chunk * 16 * NT + 16 * cb + (lane & 15);
        float val = 0.f;
        if (cin < Cin && co < Cout) {
            const float c0 = xi == 0 ? 1.f : (xi == 3 ? 0.f : .5f);
            const float c1 = xi == 1 ? .5f : (xi == 2 ? -.5f : 0.f);
            const float c2 = xi == 3 ? 1.f : (xi == 0 ? 0.f : .5f);
            const size_t tap = (size_t)Cin * Cout;
            const float* p = w + (size_t)cin * Cout + co;
            float t[3];
#pragma unroll
            for (int b = 0; b < 3; ++b) t[b] = c0 * p[(0 * 3 + b) * tap] + c1 * p[(1 * 3 + b) * tap] + c2 * p[(2 * 3 + b) * tap];
            val = nu == 0 ? t[0] : (nu == 1 ? .5f * (t[0] + t[1] + t[2]) : (nu == 2 ? .5f * (t[0] - t[1] + t[2]) : -t[2]));
        }
        u[idx] = val;
    }
}

struct WinoFilterEntry {
    const float* w; int Cin, Cout, KQ, NT, nchunk; int total; int kind;       // kind: pass it was registered in (0 forward, 1 backward)
    hipStream_t stream; float* u; bool fresh; int f44;
};
std::vector<WinoFilterEntry>& wino_entries() { static std::vector<WinoFilterEntry> v; return v; }
int g_wino_pass_depth = 0, g_wino_pass_kind = 0;
constexpr size_t WINO_ENTRIES_MAX = 512;

// -> the transformed filter to use and whether it still has to be computed (by the caller, on s)
float* wino_filter_lookup(hipStream_t s, const float* w, int Cin, int Cout, int KQ, int NT, int nchunk, int total, bool& need, int f44 = 0) {
    static const bool off = exp_env("DL4DS_WINO_NO_FILTER_CACHE") != nullptr;          // (A/B)
    need = true;
    if (g_wino_pass_depth <= 0 || off) return wino_scratch(s, (size_t)total);
    auto& es = wino_entries();
    for (auto& e : es)
        if (e.w == w && e.Cin == Cin && e.Cout == Cout && e.KQ == KQ && e.NT == NT && e.nchunk == nchunk && e.stream == s && e.f44 == f44) {
            need = !e.fresh;
            e.fresh = true;                      // (the caller transforms it now if it was not)
            return e.u;
        }
    if (es.size() >= WINO_ENTRIES_MAX) return wino_scratch(s, (size_t)total);
    WinoFilterEntry e{w, Cin, Cout, KQ, NT, nchunk, total, g_wino_pass_kind, s, nullptr, true, f44};
    HIP_CHECK(hipMalloc((void**)&e.u, (size_t)total * sizeof(float)));
    es.push_back(e);
    return e.u;
}

}  // namespace

WinoPassGuard::WinoPassGuard(int kind) : prev_kind(g_wino_pass_kind) { ++g_wino_pass_depth; g_wino_pass_kind = kind; }
WinoPassGuard::~WinoPassGuard() { --g_wino_pass_depth; g_wino_pass_kind = prev_kind; }

bool wino_pass_active(int& kind) { kind = g_wino_pass_kind; return g_wino_pass_depth > 0; }

void wino_filters_invalidate(const float* lo, const float* hi) {
    for (auto& e : wino_entries())
        if (e.w >= lo && e.w < hi) e.fresh = false;
    split_filters_invalidate(lo, hi);                      // (conv_split.hip: the six-term kernel's filter fragments follow the same life cycle)
}

void wino_filters_release(const float* lo, const float* hi) {
    split_filters_release(lo, hi);
    auto& es = wino_entries();
    for (size_t i = 0; i < es.size();) {
        if (es[i].w >= lo && es[i].w < hi) { (void)hipFree(es[i].u); es[i] = es.back(); es.pop_back(); }
        else ++i;
    }
}

void wino_filters_refresh(hipStream_t s, const float* lo, const float* hi, int kind) {
    WinoFilterJobs jobs;
    jobs.n = 0; jobs.total = 0;
    auto flush = [&]() {
        if (!jobs.n) return;
        ProfScope ps(s, "wino_filters", 0.0, 4.0 * jobs.total * WINO_FILTER_PER_BLOCK);
        DL4DS_LAUNCH(wino_filter_batched_kernel, dim3(jobs.total), dim3(256), 0, s, jobs);
        HIP_CHECK(hipGetLastError());
        jobs.n = 0; jobs.total = 0;
    };
    for (auto& e : wino_entries()) {
        if (e.fresh || e.kind != kind || e.stream != s || e.w < lo || e.w >= hi) continue;
        if (jobs.n == WINO_JOBS_MAX) flush();
        jobs.j[jobs.n++] = WinoFilterJob{e.w, e.u, e.Cin, e.Cout, e.KQ, e.NT, e.nchunk, e.total, jobs.total, e.f44};
        jobs.total += cdiv(e.total, WINO_FILTER_PER_BLOCK);
        e.fresh = true;
    }
    flush();
    split_filters_refresh(s, lo, hi, kind);
}

namespace {
}  // namespace

// Winograd F(4x4, 3x3) (conv_wino4_kernel.h): the layers whose transformed filter fits the register file of one workgroup per CU --
// cout chunks of 32 (NT = 2) with 48 or 32 input channels per pass.  Same contract as conv2d_wino_forward below, which tries this first.
static bool conv2d_wino4_forward(hipStream_t s, const TView& in, const float* w, const TView& out, const ConvEpilogue& ep, int KQ, int cpass,
                                 const char* force) {
    const char* f44 = test_env("DL4DS_WINO_F44");                 // (tests: "force" = also couts that are not whole chunks of 32)
    const bool forced = f44 && f44[0] == 'f';
    if (!forced && (out.C % 32) != 0) return false;
    if (!forced && in.C % cpass != 0 && in.C > cpass) return false;
    const int NT = 2;
    const int passes = cdiv(in.C, cpass);
    const int full_epi = (ep.add.p ? WINO_ADD : 0) | (ep.mask.p ? WINO_MASK : 0) | (ep.accumulate ? WINO_OLDA : 0);
    auto span = [](const TView& v) { const size_t r = std::max(v.d2s, 1); return (size_t)24 * v.W * r * r * v.ld * 4; };
    if (span(in) >= (1ull << 31) || span(out) >= (1ull << 31)) return false;
    WinoParams wp;
    ConvParams& p = wp.c;
    p.in = in; p.out = out; p.add = ep.add; p.mask = ep.mask;
    p.w = w; p.bias = ep.bias;
    p.Cin = in.C; p.Cout = out.C; p.H = in.H; p.W = in.W;
    p.relu = ep.relu; p.accumulate = ep.accumulate;
    wp.nchunk = cdiv(out.C, 16 * NT);
    wp.tgx = cdiv(in.W, 16);
    wp.tgy = cdiv(in.H, 16);
    wp.m_tgx = div_magic(wp.tgx);
    wp.m_tgy = div_magic(wp.tgy);
    const long ntg = (long)wp.tgx * wp.tgy * in.N;
    if (ntg >= (1l << 20)) return false;
    wp.ntg = (int)ntg;
    wp.per_xcd = cdiv(wp.ntg, 8);
    const int SXmax = std::max(wino_cu_count() / 8, 1);                        // one workgroup per CU
    if (wp.nchunk > SXmax) return false;
    int SX = (SXmax / wp.nchunk) * wp.nchunk;
    if (force && atoi(force) > 0) SX = std::min(SX, atoi(force) * wp.nchunk);
    if (!force && (long)wp.per_xcd * wp.nchunk < 4l * SX) return false;       // fewer than four tile groups per workgroup
    const double px = (double)in.N * in.H * in.W;
    // (issued work: 36 multiply-adds per 4x4 tile and (cin, cout) pair of the padded operands, plus the transforms' operations)
    const double issued = 2.0 * (px / 16.0) * 36.0 * (16.0 * KQ * passes) * (16.0 * NT * wp.nchunk) +
                          (px / 16.0) * (2.0 * 154.0 * 4.0 * in.C * wp.nchunk / 4.0 + 124.0 * out.C * passes);
    ProfScope ps(s, "conv_wino4<" + std::to_string(KQ) + "," + std::to_string(NT) + ">", issued,
                 4.0 * (px * (in.C + out.C * (1 + (ep.add.p ? 1 : 0) + (ep.mask.p ? 1 : 0) + (ep.accumulate ? 1 : 0))) + 9.0 * in.C * out.C),
                 2.0 * px * 9.0 * in.C * out.C);
    const size_t per_pass = (size_t)wp.nchunk * 4 * (36 * KQ * NT) * 64;
    const int total = (int)(per_pass * passes);
    bool need = true;
    float* const u = wino_filter_lookup(s, w, in.C, out.C, KQ, NT, wp.nchunk, total, need, 1);
    if (need) {
        DL4DS_LAUNCH(wino4_filter_kernel, dim3(std::min(cdiv(total, 256), 2048)), dim3(256), 0, s, w, u, in.C, out.C, KQ, NT,
                     wp.nchunk, total);
        HIP_CHECK(hipGetLastError());
    }
    for (int ps_ = 0; ps_ < passes; ++ps_) {
        const bool last = ps_ == passes - 1;
        wp.cin0 = ps_ * cpass;
        wp.u = u + per_pass * ps_;
        wp.first = ps_ == 0;
        p.relu = last ? ep.relu : 0;
        const int epi = passes == 1 ? full_epi : ((ps_ ? WINO_OLDF : 0) | (last ? full_epi : 0));
        if (KQ == 2) launch_wino4_22(s, wp, SX, epi); else launch_wino4_32(s, wp, SX, epi);
    }
    return true;
}

// 3x3, stride 1, SAME.  Returns false when the layer is not eligible (the caller falls through to the direct kernels).
bool conv2d_wino_forward(hipStream_t s, const TView& in, const float* w, const TView& out, const ConvEpilogue& ep) {
    const bool off = getenv("DL4DS_NO_WINOGRAD") != nullptr;
    const char* force = test_env("DL4DS_WINO_FORCE");          // (tests: small grids too; "<k>" = k workgroups per XCD and cout chunk)
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
    const int full_epi = (ep.add.p ? WINO_ADD : 0) | (ep.mask.p ? WINO_MASK : 0) | (ep.accumulate ? WINO_OLDA : 0);
    if (!wino_epi_built(full_epi) || (passes > 1 && !wino_epi_built(full_epi | WINO_OLDF))) return false;
    if (test_env("DL4DS_WINO_F44") && conv2d_wino4_forward(s, in, w, out, ep, KQ, cpass, force)) return true;
    // cout chunks of 32 or 48: the least padding, then the wider chunk (the input transform is paid once per chunk)
    const int NT = (cdiv(out.C, 32) * 32 < cdiv(out.C, 48) * 48) ? 2 : 3;
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
    const int SXmax = std::max(2 * wino_cu_count() / 8, 1);                    // two workgroups per CU
    if (wp.nchunk > SXmax) return false;
    int SX = (SXmax / wp.nchunk) * wp.nchunk;
    if (force && atoi(force) > 0) SX = std::min(SX, atoi(force) * wp.nchunk);
    if (!force && (long)wp.per_xcd * wp.nchunk < 4l * SX) return false;       // fewer than four tile groups per workgroup
    const double px = (double)in.N * in.H * in.W;
    // (issued work: 16 multiply-adds per 2x2 tile and (cin, cout) pair of the padded operands, plus the transforms' additions)
    const double issued = 2.0 * (px / 4.0) * 16.0 * (16.0 * KQ * passes) * (16.0 * NT * wp.nchunk) +
                          (px / 4.0) * (32.0 * in.C * wp.nchunk + 24.0 * out.C * passes);
    ProfScope ps(s, "conv_wino<" + std::to_string(KQ) + "," + std::to_string(NT) + ">", issued,
                 4.0 * (px * (in.C + out.C * (1 + (ep.add.p ? 1 : 0) + (ep.mask.p ? 1 : 0) + (ep.accumulate ? 1 : 0))) + 9.0 * in.C * out.C),
                 2.0 * px * 9.0 * in.C * out.C);
    const size_t per_pass = (size_t)wp.nchunk * 4 * (16 * KQ * NT) * 64;
    const int total = (int)(per_pass * passes);
    bool need = true;
    float* const u = wino_filter_lookup(s, w, in.C, out.C, KQ, NT, wp.nchunk, total, need);
    if (need) {
        DL4DS_LAUNCH(wino_filter_kernel, dim3(std::min(cdiv(total, 256), 2048)), dim3(256), 0, s, w, u, in.C, out.C, KQ, NT,
                           wp.nchunk, total);
        HIP_CHECK(hipGetLastError());
    }
    for (int ps_ = 0; ps_ < passes; ++ps_) {
        const bool last = ps_ == passes - 1;
        wp.cin0 = ps_ * cpass;
        wp.u = u + per_pass * ps_;
        wp.first = ps_ == 0;
        p.relu = last ? ep.relu : 0;
        const int epi = passes == 1 ? full_epi : ((ps_ ? WINO_OLDF : 0) | (last ? full_epi : 0));
        if (KQ == 2) {
            if (NT == 2) launch_wino_22(s, wp, SX, epi); else launch_wino_23(s, wp, SX, epi);
        } else {
            if (NT == 2) launch_wino_32(s, wp, SX, epi); else launch_wino_33(s, wp, SX, epi);
        }
    }
    return true;
}
