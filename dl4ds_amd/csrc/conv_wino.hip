// dl4ds_amd -- Winograd F(2x2, 3x3) convolution: eligibility, channel passes, epilogue form (kernel: conv_wino_kernel.h)
#include "conv_wino_kernel.h"

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
// second form of the kernel (conv_wino2_kernel): slots 1 and 2 hold U1 - U2 and U1 + U2 (see its K loop)
__global__ void __launch_bounds__(256) wino_filter_kernel(const float* __restrict__ w, float* __restrict__ u, int Cin, int Cout, int KQ,
                                                          int NT, int nchunk, int total, int form2) {
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
            if (form2) val = nu == 0 ? t[0] : (nu == 1 ? t[1] : (nu == 2 ? t[0] + t[2] : -t[2]));
            else val = nu == 0 ? t[0] : (nu == 1 ? .5f * (t[0] + t[1] + t[2]) : (nu == 2 ? .5f * (t[0] - t[1] + t[2]) : -t[2]));
        }
        u[idx] = val;
    }
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
    const int full_epi = (ep.add.p ? WINO_ADD : 0) | (ep.mask.p ? WINO_MASK : 0) | (ep.accumulate ? WINO_OLDA : 0);
    if (!wino_epi_built(full_epi) || (passes > 1 && !wino_epi_built(full_epi | WINO_OLDF))) return false;
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
    float* const u = wino_scratch(s, per_pass * passes);
    {
        const int total = (int)(per_pass * passes);
        DL4DS_LAUNCH(wino_filter_kernel, dim3(std::min(cdiv(total, 256), 2048)), dim3(256), 0, s, w, u, in.C, out.C, KQ, NT,
                           wp.nchunk, total, 0);
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
