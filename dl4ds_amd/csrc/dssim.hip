// DSSIM loss of dl4ds/losses.py:23-55 fused with its gradient w.r.t. the prediction:
//   drange = max(max y, max p) - min(min y, min p)   (whole batch; differentiated through, like TF does)
//   x = y - min(y) if min(y) < 0 else y ;  q = p - min(p) if min(p) < 0 else p
//   ssim   = tf.image.ssim(x, q, max_val=drange, 11x11 gaussian sigma 1.5, VALID, k1=.01, k2=.03)
//   dssim  = mean_n((1 - ssim_n)/2)
// Kernels: (1) min/max(+arg) reduction, (2) 16x16-tile SSIM map from an LDS halo tile -- the 11x11 Gaussian is applied
// separably (row pass over the 26 halo rows into LDS, then the column pass per output: 2 x 11 taps instead of 121) to the
// four moments -- emitting the three per-map derivatives, (3) the transposed (again separable) filter of those
// derivatives per input pixel, (4) scalar fix-ups (drange and shift terms routed to arg-max / arg-min).
#include "ops.h"
#include "prof.h"
#include <algorithm>

namespace {

constexpr int KF = 11, KH = 5, TS = 16, TL = TS + KF - 1;    // filter, half, tile, tile+halo
constexpr int PT = 48;      // LDS pitch of the halo tiles: consecutive rows start 16 banks apart, so the 2 rows x 16 columns
                            // a 32-lane LDS pass touches are conflict-free (the row-filtered tiles use pitch TS = 16 likewise)

struct Gauss { float g[KF]; };
Gauss make_gauss() {
    Gauss k;
    double s = 0;
    for (int i = 0; i < KF; ++i) { double c = i - (KF - 1) / 2.0; k.g[i] = (float)std::exp(-c * c / (2.0 * 1.5 * 1.5)); s += k.g[i]; }
    for (int i = 0; i < KF; ++i) k.g[i] = (float)(k.g[i] / s);   // outer(g,g)/sum == outer(g/s, g/s)
    return k;
}

struct Stats {            // device-resident scalars
    float minT, maxT, minP, maxP;
    unsigned long long argminP, argmaxP;
    float sumS, gc1, gc2, sumdy;
};

__global__ void __launch_bounds__(256) minmax_kernel(const float* __restrict__ t, const float* __restrict__ p, size_t n,
                                                     float* __restrict__ pf, unsigned long long* __restrict__ pi) {
    __shared__ float s[4][256];
    __shared__ unsigned long long si[2][256];
    float mnT = 3.4e38f, mxT = -3.4e38f, mnP = 3.4e38f, mxP = -3.4e38f;
    unsigned long long amn = 0, amx = 0;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const float a = t[e], b = p[e];
        mnT = fminf(mnT, a); mxT = fmaxf(mxT, a);
        if (b < mnP) { mnP = b; amn = e; }
        if (b > mxP) { mxP = b; amx = e; }
    }
    const int i = threadIdx.x;
    s[0][i] = mnT; s[1][i] = mxT; s[2][i] = mnP; s[3][i] = mxP; si[0][i] = amn; si[1][i] = amx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (i < o) {
            s[0][i] = fminf(s[0][i], s[0][i + o]);
            s[1][i] = fmaxf(s[1][i], s[1][i + o]);
            if (s[2][i + o] < s[2][i] || (s[2][i + o] == s[2][i] && si[0][i + o] < si[0][i])) { s[2][i] = s[2][i + o]; si[0][i] = si[0][i + o]; }
            if (s[3][i + o] > s[3][i] || (s[3][i + o] == s[3][i] && si[1][i + o] < si[1][i])) { s[3][i] = s[3][i + o]; si[1][i] = si[1][i + o]; }
        }
        __syncthreads();
    }
    if (i == 0) {
        for (int k = 0; k < 4; ++k) pf[blockIdx.x * 4 + k] = s[k][0];
        pi[blockIdx.x * 2] = si[0][0];
        pi[blockIdx.x * 2 + 1] = si[1][0];
    }
}
__global__ void __launch_bounds__(256) minmax_finish_kernel(const float* __restrict__ pf, const unsigned long long* __restrict__ pi,
                                                            int nb, Stats* st) {
    __shared__ float s[4][256];
    __shared__ unsigned long long si[2][256];
    const int i = threadIdx.x;
    float mnT = 3.4e38f, mxT = -3.4e38f, mnP = 3.4e38f, mxP = -3.4e38f;
    unsigned long long amn = ~0ull, amx = ~0ull;
    for (int k = i; k < nb; k += 256) {
        mnT = fminf(mnT, pf[4 * k]); mxT = fmaxf(mxT, pf[4 * k + 1]);
        if (pf[4 * k + 2] < mnP || (pf[4 * k + 2] == mnP && pi[2 * k] < amn)) { mnP = pf[4 * k + 2]; amn = pi[2 * k]; }
        if (pf[4 * k + 3] > mxP || (pf[4 * k + 3] == mxP && pi[2 * k + 1] < amx)) { mxP = pf[4 * k + 3]; amx = pi[2 * k + 1]; }
    }
    s[0][i] = mnT; s[1][i] = mxT; s[2][i] = mnP; s[3][i] = mxP; si[0][i] = amn; si[1][i] = amx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {          // ties go to the smallest flat index, as a serial scan would pick
        if (i < o) {
            s[0][i] = fminf(s[0][i], s[0][i + o]);
            s[1][i] = fmaxf(s[1][i], s[1][i + o]);
            if (s[2][i + o] < s[2][i] || (s[2][i + o] == s[2][i] && si[0][i + o] < si[0][i])) { s[2][i] = s[2][i + o]; si[0][i] = si[0][i + o]; }
            if (s[3][i + o] > s[3][i] || (s[3][i + o] == s[3][i] && si[1][i + o] < si[1][i])) { s[3][i] = s[3][i + o]; si[1][i] = si[1][i + o]; }
        }
        __syncthreads();
    }
    if (i) return;
    st->minT = s[0][0]; st->maxT = s[1][0]; st->minP = s[2][0]; st->maxP = s[3][0]; st->argminP = si[0][0]; st->argmaxP = si[1][0];
    st->sumS = 0.f; st->gc1 = 0.f; st->gc2 = 0.f; st->sumdy = 0.f;
}

// Gaussian-filtered moments (mean x, mean q, E[xq], E[x^2 + q^2]) of this thread's output pixel of the tile: row pass of
// all TL halo rows into h (all 256 threads; contains a barrier), then the column pass.
struct MsMoments { float mx, my, A, Bq; };
__device__ __forceinline__ MsMoments sep_moments(const float (*sx)[PT], const float (*sq)[PT], float (*h)[TL][TS], const Gauss& gk) {
    for (int it = threadIdx.x; it < TL * TS; it += 256) {
        const int r = it / TS, cx = it % TS;
        float rx = 0.f, ry = 0.f, ra = 0.f, rb = 0.f;
#pragma unroll
        for (int j = 0; j < KF; ++j) {
            const float a = sx[r][cx + j], q = sq[r][cx + j], w = gk.g[j];
            rx += w * a; ry += w * q; ra += w * a * q; rb += w * (a * a + q * q);
        }
        h[0][r][cx] = rx; h[1][r][cx] = ry; h[2][r][cx] = ra; h[3][r][cx] = rb;
    }
    __syncthreads();
    const int ly = threadIdx.x / TS, lx = threadIdx.x % TS;
    MsMoments m{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < KF; ++i) {
        const float w = gk.g[i];
        m.mx += w * h[0][ly + i][lx]; m.my += w * h[1][ly + i][lx]; m.A += w * h[2][ly + i][lx]; m.Bq += w * h[3][ly + i][lx];
    }
    return m;
}
// transposed filter of three derivative maps held as halo tiles starting (KF-1) pixels before the tile:
// T_k = sum_{i,j} g[i] g[j] s_k[ly + KF-1 - i][lx + KF-1 - j]
__device__ __forceinline__ void sep_transposed3(const float (*s1)[PT], const float (*s2)[PT], const float (*s3)[PT],
                                                float (*h)[TL][TS], const Gauss& gk, float& T1, float& T2, float& T3) {
    for (int it = threadIdx.x; it < TL * TS; it += 256) {
        const int r = it / TS, cx = it % TS;
        float r1 = 0.f, r2 = 0.f, r3 = 0.f;
#pragma unroll
        for (int j = 0; j < KF; ++j) {
            const float w = gk.g[j];
            r1 += w * s1[r][cx + (KF - 1) - j]; r2 += w * s2[r][cx + (KF - 1) - j]; r3 += w * s3[r][cx + (KF - 1) - j];
        }
        h[0][r][cx] = r1; h[1][r][cx] = r2; h[2][r][cx] = r3;
    }
    __syncthreads();
    const int ly = threadIdx.x / TS, lx = threadIdx.x % TS;
    T1 = T2 = T3 = 0.f;
#pragma unroll
    for (int i = 0; i < KF; ++i) {
        const float w = gk.g[i];
        T1 += w * h[0][ly + (KF - 1) - i][lx]; T2 += w * h[1][ly + (KF - 1) - i][lx]; T3 += w * h[2][ly + (KF - 1) - i][lx];
    }
}

// one block = one 16x16 tile of the (Ho,Wo) SSIM map of plane (n,c)
__global__ void __launch_bounds__(256) ssim_fwd_kernel(const float* __restrict__ t, const float* __restrict__ p, int H, int W,
                                                       int C, int Ho, int Wo, int tiles_x, int tiles_y, Gauss gk,
                                                       const Stats* __restrict__ st, float* __restrict__ dmu,
                                                       float* __restrict__ da, float* __restrict__ db,
                                                       float* __restrict__ partial) {
    __shared__ float sx[TL][PT], sq[TL][PT], hrow[4][TL][TS];
    __shared__ float red[3][256];
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y; b /= tiles_y;
    const int c = b % C, n = b / C;
    const float shT = st->minT < 0.f ? st->minT : 0.f, shP = st->minP < 0.f ? st->minP : 0.f;
    const float drange = fmaxf(st->maxT, st->maxP) - fminf(st->minT, st->minP);
    const float c1 = (0.01f * drange) * (0.01f * drange), c2 = (0.03f * drange) * (0.03f * drange);
    const int oy0 = ty * TS, ox0 = tx * TS;
    for (int i = threadIdx.x; i < TL * TL; i += 256) {
        const int r = i / TL, cc = i % TL;
        const int y = oy0 + r, x = ox0 + cc;
        float a = 0.f, q = 0.f;
        if (y < H && x < W) {
            const size_t o = (((size_t)n * H + y) * W + x) * C + c;
            a = t[o] - shT; q = p[o] - shP;
        }
        sx[r][cc] = a; sq[r][cc] = q;
    }
    __syncthreads();
    const int ly = threadIdx.x / TS, lx = threadIdx.x % TS;
    const int oy = oy0 + ly, ox = ox0 + lx;
    float S = 0.f, g1 = 0.f, g2 = 0.f;
    const MsMoments mom = sep_moments(sx, sq, hrow, gk);
    if (oy < Ho && ox < Wo) {
        const float mx = mom.mx, my = mom.my, A = mom.A, Bq = mom.Bq;
        const float N1 = 2.f * mx * my + c1, D1 = mx * mx + my * my + c1;
        const float N2 = 2.f * A - 2.f * mx * my + c2, D2 = Bq - mx * mx - my * my + c2;
        const float lum = N1 / D1, cs = N2 / D2;
        S = lum * cs;
        const size_t o = (((size_t)n * C + c) * Ho + oy) * Wo + ox;
        dmu[o] = cs * (2.f * mx / D1 - N1 * 2.f * my / (D1 * D1)) + lum * (-2.f * mx / D2 + N2 * 2.f * my / (D2 * D2));
        da[o] = lum * 2.f / D2;
        db[o] = -lum * N2 / (D2 * D2);
        g1 = cs * (D1 - N1) / (D1 * D1);
        g2 = lum * (D2 - N2) / (D2 * D2);
    }
    red[0][threadIdx.x] = S; red[1][threadIdx.x] = g1; red[2][threadIdx.x] = g2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        for (int k = 0; k < 3; ++k) partial[(size_t)blockIdx.x * 3 + k] = red[k][0];
}

// First stage of the long partial-sum reductions: out[g][k] = sum of in[r][k] over the g-th slice of the nb rows (fixed
// order, double accumulation), so the single-block finishing kernels read COMPACT_G rows instead of one per tile.
constexpr int COMPACT_G = 64;
template <int K>
__global__ void __launch_bounds__(256) compact_partials_kernel(const float* __restrict__ in, int nb, float* __restrict__ out) {
    __shared__ double red[K][256];
    const int chunk = (nb + (int)gridDim.x - 1) / (int)gridDim.x;
    const int r0 = blockIdx.x * chunk, r1 = min(r0 + chunk, nb);
    double a[K];
#pragma unroll
    for (int k = 0; k < K; ++k) a[k] = 0.0;
    for (int r = r0 + threadIdx.x; r < r1; r += 256)
#pragma unroll
        for (int k = 0; k < K; ++k) a[k] += (double)in[(size_t)r * K + k];
#pragma unroll
    for (int k = 0; k < K; ++k) red[k][threadIdx.x] = a[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
#pragma unroll
            for (int k = 0; k < K; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < K; ++k) out[(size_t)blockIdx.x * K + k] = (float)red[k][0];
}

__global__ void sum3_kernel(const float* __restrict__ partial, int nb, Stats* st) {
    __shared__ double red[3][256];
    double a[3] = {0, 0, 0};
    for (int k = threadIdx.x; k < nb; k += 256)
        for (int j = 0; j < 3; ++j) a[j] += partial[(size_t)k * 3 + j];
    for (int j = 0; j < 3; ++j) red[j][threadIdx.x] = a[j];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            for (int j = 0; j < 3; ++j) red[j][threadIdx.x] += red[j][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) { st->sumS = (float)red[0][0]; st->gc1 = (float)red[1][0]; st->gc2 = (float)red[2][0]; }
}

// one block = one 16x16 tile of INPUT pixels of plane (n,c): transposed 11x11 filter of the derivative maps
__global__ void __launch_bounds__(256) ssim_bwd_kernel(const float* __restrict__ t, const float* __restrict__ p, int H, int W,
                                                       int C, int Ho, int Wo, int tiles_x, int tiles_y, Gauss gk,
                                                       const Stats* __restrict__ st, const float* __restrict__ dmu,
                                                       const float* __restrict__ da, const float* __restrict__ db,
                                                       float coef, float* __restrict__ dpred, int accumulate,
                                                       float* __restrict__ partial) {
    __shared__ float s1[TL][PT], s2[TL][PT], s3[TL][PT], hrow[3][TL][TS];
    __shared__ float red[256];
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y; b /= tiles_y;
    const int c = b % C, n = b / C;
    const float shT = st->minT < 0.f ? st->minT : 0.f, shP = st->minP < 0.f ? st->minP : 0.f;
    const int y0 = ty * TS, x0 = tx * TS;
    // input pixel (y,x) is covered by outputs (y-i, x-j), i,j in [0,10]: halo tile starts at (y0-10, x0-10)
    for (int i = threadIdx.x; i < TL * TL; i += 256) {
        const int r = i / TL, cc = i % TL;
        const int oy = y0 - (KF - 1) + r, ox = x0 - (KF - 1) + cc;
        float a = 0.f, bb = 0.f, d = 0.f;
        if (oy >= 0 && oy < Ho && ox >= 0 && ox < Wo) {
            const size_t o = (((size_t)n * C + c) * Ho + oy) * Wo + ox;
            a = dmu[o]; bb = da[o]; d = db[o];
        }
        s1[r][cc] = a; s2[r][cc] = bb; s3[r][cc] = d;
    }
    __syncthreads();
    const int ly = threadIdx.x / TS, lx = threadIdx.x % TS;
    const int y = y0 + ly, x = x0 + lx;
    float g = 0.f;
    float T1, T2, T3;
    sep_transposed3(s1, s2, s3, hrow, gk, T1, T2, T3);
    if (y < H && x < W) {
        const size_t o = (((size_t)n * H + y) * W + x) * C + c;
        const float xv = t[o] - shT, qv = p[o] - shP;
        g = coef * (T1 + xv * T2 + 2.f * qv * T3);
        dpred[o] = accumulate ? dpred[o] + g : g;
    }
    red[threadIdx.x] = g;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ void dssim_finish_kernel(const float* __restrict__ partial, int nb, Stats* st, float weight, float inv_m,
                                    float coef, float* __restrict__ dpred, float* __restrict__ loss_out, int accumulate_loss) {
    __shared__ double red[256];
    double a = 0;
    for (int k = threadIdx.x; k < nb; k += 256) a += partial[k];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    const float sumdy = (float)red[0];
    const float v = weight * (0.5f - 0.5f * st->sumS * inv_m);
    loss_out[0] = accumulate_loss ? loss_out[0] + v : v;
    if (!dpred) return;
    const float drange = fmaxf(st->maxT, st->maxP) - fminf(st->minT, st->minP);
    // dL/ddrange through c1=(k1*L)^2, c2=(k2*L)^2
    const float ddr = coef * (st->gc1 * 2.f * 0.01f * 0.01f * drange + st->gc2 * 2.f * 0.03f * 0.03f * drange);
    if (st->maxP > st->maxT) dpred[st->argmaxP] += ddr;      // tf.maximum routes ties to its first argument (y_true)
    if (st->minP < st->minT) dpred[st->argminP] -= ddr;
    if (st->minP < 0.f) dpred[st->argminP] -= sumdy;         // q = p - min(p)
}

struct Layout { size_t stats, pf, pi, maps, part, part2, total; int nb_mm; };
Layout layout(int N, int H, int W, int C) {
    Layout l;
    const int Ho = H - KF + 1, Wo = W - KF + 1;
    l.nb_mm = 512;
    size_t o = 0;
    auto bump = [&](size_t bytes) { size_t r = o; o += (bytes + 255) & ~(size_t)255; return r; };
    l.stats = bump(sizeof(Stats));
    l.pf = bump((size_t)l.nb_mm * 4 * sizeof(float));
    l.pi = bump((size_t)l.nb_mm * 2 * sizeof(unsigned long long));
    l.maps = bump(3 * (size_t)N * C * Ho * Wo * sizeof(float));
    const size_t tiles = (size_t)N * C * cdiv(H, TS) * cdiv(W, TS);
    l.part = bump(tiles * 3 * sizeof(float));
    l.part2 = bump((size_t)COMPACT_G * 3 * sizeof(float));
    l.total = o;
    return l;
}

// ============================================================================================ multi-scale DSSIM
// msdssim of dl4ds/losses.py:92-130 (+ the mixes :133-149): tf.image.ssim_multiscale with four scales.  Scale k is the
// shifted image pair halved k times (2x2 average, odd sizes SYMMETRIC-padded = edge row/column repeated).  Per scale and
// per (n, c): cs_k = mean contrast-structure, and on the last scale ssim_3 = mean luminance*cs; relu'd,
//     ms[n,c] = cs_0^w0 * cs_1^w1 * cs_2^w2 * ssim_3^w3 ,  loss = weight * mean_n (1 - mean_c ms) / 2 .
// Passes: pooled pyramids -> per-tile (ssim, cs) sums per scale -> per-(n,c) means -> one block combines them into the
// loss and the upstream weights G -> per scale: derivative maps (moments recomputed, weighted by G) -> transposed filter
// into a per-scale image gradient -> coarse-to-fine accumulation through the pooling adjoint -> dpred, plus the same
// drange / min-shift fix-ups as the single-scale loss.  All reductions have a fixed order (deterministic).
constexpr int MS_SCALES = 4;
struct MsDims { int H[MS_SCALES], W[MS_SCALES]; };

// dst (Hd, Wd) = 2x2 average of src (Hs, Ws) minus `shift` (scale 0 -> 1 applies the positivity shift), edge repeated
__global__ void ms_pool_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int Hs, int Ws, int Hd, int Wd,
                               int C, const Stats* __restrict__ st, int which) {      // which: 0 none, 1 y_true, 2 y_pred
    const float sh = which == 1 ? (st->minT < 0.f ? st->minT : 0.f) : which == 2 ? (st->minP < 0.f ? st->minP : 0.f) : 0.f;
    const size_t total = (size_t)N * Hd * Wd * C;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        size_t r = e / C;
        const int x = (int)(r % Wd); r /= Wd;
        const int y = (int)(r % Hd);
        const int n = (int)(r / Hd);
        const int y0 = 2 * y, y1 = min(2 * y + 1, Hs - 1), x0 = 2 * x, x1 = min(2 * x + 1, Ws - 1);
        const float* b = src + (size_t)n * Hs * Ws * C + c;
        const float v = b[((size_t)y0 * Ws + x0) * C] + b[((size_t)y0 * Ws + x1) * C] + b[((size_t)y1 * Ws + x0) * C] +
                        b[((size_t)y1 * Ws + x1) * C];
        dst[e] = 0.25f * v - sh;
    }
}

// fine (Hs, Ws) += pooling adjoint of coarse (Hd, Wd): each of the four (possibly repeated) sources gets 1/4
__global__ void ms_unpool_add_kernel(float* __restrict__ fine, const float* __restrict__ coarse, int N, int Hs, int Ws, int Hd,
                                     int Wd, int C) {
    const size_t total = (size_t)N * Hs * Ws * C;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        size_t r = e / C;
        const int x = (int)(r % Ws); r /= Ws;
        const int y = (int)(r % Hs);
        const int n = (int)(r / Hs);
        const float wy = ((Hs & 1) && y == Hs - 1) ? 2.f : 1.f;      // the repeated edge row counts twice
        const float wx = ((Ws & 1) && x == Ws - 1) ? 2.f : 1.f;
        fine[e] += 0.25f * wy * wx * coarse[(((size_t)n * Hd + y / 2) * Wd + x / 2) * C + c];
    }
}

// MODE 0: partial[block] = (sum ssim, sum cs) of the tile.  MODE 1: derivative maps weighted by gs[n,c] (ssim) and gc[n,c]
// (cs), partial[block] = (d/dc1, d/dc2).  x / q: scale images (scale 0: raw y_true / y_pred, shifted here).
template <int MODE>
__global__ void __launch_bounds__(256) ms_ssim_kernel(const float* __restrict__ x, const float* __restrict__ q, int raw, int H, int W,
                                                      int C, int Ho, int Wo, int tiles_x, int tiles_y, Gauss gk,
                                                      const Stats* __restrict__ st, const float* __restrict__ gs,
                                                      const float* __restrict__ gc, float* __restrict__ dmu, float* __restrict__ da,
                                                      float* __restrict__ db, float* __restrict__ partial) {
    __shared__ float sx[TL][PT], sq[TL][PT], hrow[4][TL][TS];
    __shared__ float red[2][256];
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y; b /= tiles_y;
    const int c = b % C, n = b / C;
    const float shT = (raw && st->minT < 0.f) ? st->minT : 0.f, shP = (raw && st->minP < 0.f) ? st->minP : 0.f;
    const float drange = fmaxf(st->maxT, st->maxP) - fminf(st->minT, st->minP);
    const float c1 = (0.01f * drange) * (0.01f * drange), c2 = (0.03f * drange) * (0.03f * drange);
    const int oy0 = ty * TS, ox0 = tx * TS;
    for (int i = threadIdx.x; i < TL * TL; i += 256) {
        const int r = i / TL, cc = i % TL;
        const int y = oy0 + r, xx = ox0 + cc;
        float a = 0.f, v = 0.f;
        if (y < H && xx < W) {
            const size_t o = (((size_t)n * H + y) * W + xx) * C + c;
            a = x[o] - shT; v = q[o] - shP;
        }
        sx[r][cc] = a; sq[r][cc] = v;
    }
    __syncthreads();
    const int ly = threadIdx.x / TS, lx = threadIdx.x % TS;
    const int oy = oy0 + ly, ox = ox0 + lx;
    float r0 = 0.f, r1 = 0.f;
    const MsMoments m = sep_moments(sx, sq, hrow, gk);
    if (oy < Ho && ox < Wo) {
        const float N1 = 2.f * m.mx * m.my + c1, D1 = m.mx * m.mx + m.my * m.my + c1;
        const float N2 = 2.f * m.A - 2.f * m.mx * m.my + c2, D2 = m.Bq - m.mx * m.mx - m.my * m.my + c2;
        const float lum = N1 / D1, cs = N2 / D2;
        if (MODE == 0) {
            r0 = lum * cs; r1 = cs;
        } else {
            const float inv_m = 1.f / ((float)Ho * (float)Wo);
            const float ws = gs[n * C + c] * inv_m, wc = gc[n * C + c] * inv_m;
            const float dcs_dmy = -2.f * m.mx / D2 + N2 * 2.f * m.my / (D2 * D2);
            const float dlum_dmy = 2.f * m.mx / D1 - N1 * 2.f * m.my / (D1 * D1);
            const size_t o = (((size_t)n * C + c) * Ho + oy) * Wo + ox;
            dmu[o] = ws * (cs * dlum_dmy + lum * dcs_dmy) + wc * dcs_dmy;
            da[o] = (ws * lum + wc) * 2.f / D2;
            db[o] = -(ws * lum + wc) * N2 / (D2 * D2);
            r0 = ws * cs * (D1 - N1) / (D1 * D1);                        // d/dc1
            r1 = (ws * lum + wc) * (D2 - N2) / (D2 * D2);                // d/dc2
        }
    }
    red[0][threadIdx.x] = r0; red[1][threadIdx.x] = r1;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[(size_t)blockIdx.x * 2] = red[0][0]; partial[(size_t)blockIdx.x * 2 + 1] = red[1][0]; }
}

// mean_ssim[nc], mean_cs[nc] of one scale from its per-tile sums (tiles of a plane are consecutive blocks); one wavefront
// per plane: lane-strided loads, fixed shuffle tree
__global__ void __launch_bounds__(64) ms_means_kernel(const float* __restrict__ partial, int tiles, int NC, float inv_m,
                                                      float* __restrict__ mean_ssim, float* __restrict__ mean_cs) {
    const int nc = blockIdx.x;
    double a = 0, b = 0;
    for (int k = threadIdx.x; k < tiles; k += 64) { a += partial[((size_t)nc * tiles + k) * 2]; b += partial[((size_t)nc * tiles + k) * 2 + 1]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    if (threadIdx.x == 0) {
        mean_ssim[nc] = (float)(a * inv_m);
        mean_cs[nc] = (float)(b * inv_m);
    }
}

// one block: loss value and the upstream weights gs[k][nc] (only the last scale), gc[k][nc] (the other scales)
__global__ void __launch_bounds__(256) ms_combine_kernel(const float* __restrict__ mean_ssim, const float* __restrict__ mean_cs, int NC,
                                                         float weight, float* __restrict__ gs, float* __restrict__ gc,
                                                         float* __restrict__ loss_out, int accumulate_loss) {
    const float pw[MS_SCALES] = {0.0448f, 0.2856f, 0.3001f, 0.2363f};
    __shared__ double red[256];
    double acc = 0;
    for (int nc = threadIdx.x; nc < NC; nc += 256) {
        float v[MS_SCALES], ms = 1.f;
#pragma unroll
        for (int k = 0; k < MS_SCALES; ++k) {
            v[k] = fmaxf(k == MS_SCALES - 1 ? mean_ssim[k * NC + nc] : mean_cs[k * NC + nc], 0.f);
            ms *= powf(v[k], pw[k]);
        }
        acc += ms;
        const float up = -0.5f * weight / (float)NC;                       // dL/dms[n,c]
#pragma unroll
        for (int k = 0; k < MS_SCALES; ++k) {
            const float g = v[k] > 0.f ? up * pw[k] * ms / v[k] : 0.f;
            gs[k * NC + nc] = (k == MS_SCALES - 1) ? g : 0.f;
            gc[k * NC + nc] = (k == MS_SCALES - 1) ? 0.f : g;
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float v = weight * (0.5f - 0.5f * (float)(red[0] / NC));
        loss_out[0] = accumulate_loss ? loss_out[0] + v : v;
    }
}

// transposed 11x11 filter of the derivative maps -> gradient w.r.t. the scale's q image (stored, not accumulated)
__global__ void __launch_bounds__(256) ms_bwd_kernel(const float* __restrict__ x, const float* __restrict__ q, int raw, int H, int W, int C,
                                                     int Ho, int Wo, int tiles_x, int tiles_y, Gauss gk, const Stats* __restrict__ st,
                                                     const float* __restrict__ dmu, const float* __restrict__ da,
                                                     const float* __restrict__ db, float* __restrict__ gq) {
    __shared__ float s1[TL][PT], s2[TL][PT], s3[TL][PT], hrow[3][TL][TS];
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y; b /= tiles_y;
    const int c = b % C, n = b / C;
    const float shT = (raw && st->minT < 0.f) ? st->minT : 0.f, shP = (raw && st->minP < 0.f) ? st->minP : 0.f;
    const int y0 = ty * TS, x0 = tx * TS;
    for (int i = threadIdx.x; i < TL * TL; i += 256) {
        const int r = i / TL, cc = i % TL;
        const int oy = y0 - (KF - 1) + r, ox = x0 - (KF - 1) + cc;
        float a = 0.f, bb = 0.f, d = 0.f;
        if (oy >= 0 && oy < Ho && ox >= 0 && ox < Wo) {
            const size_t o = (((size_t)n * C + c) * Ho + oy) * Wo + ox;
            a = dmu[o]; bb = da[o]; d = db[o];
        }
        s1[r][cc] = a; s2[r][cc] = bb; s3[r][cc] = d;
    }
    __syncthreads();
    const int ly = threadIdx.x / TS, lx = threadIdx.x % TS;
    const int y = y0 + ly, xx = x0 + lx;
    float T1, T2, T3;
    sep_transposed3(s1, s2, s3, hrow, gk, T1, T2, T3);
    if (y < H && xx < W) {
        const size_t o = (((size_t)n * H + y) * W + xx) * C + c;
        gq[o] = T1 + (x[o] - shT) * T2 + 2.f * (q[o] - shP) * T3;
    }
}

// dpred (+)= g0 ; partial[block] = sum of g0 over the block's elements (for the min-shift term)
__global__ void __launch_bounds__(256) ms_apply_kernel(const float* __restrict__ g0, float* __restrict__ dpred, size_t n, int accumulate,
                                                       float* __restrict__ partial) {
    __shared__ float red[256];
    float s = 0.f;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const float g = g0[e];
        dpred[e] = accumulate ? dpred[e] + g : g;
        s += g;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// scalar fix-ups: drange enters every scale's c1, c2; the prediction's minimum shifts the whole image
__global__ void __launch_bounds__(256) ms_finish_kernel(const float* __restrict__ sum_partial, int nb_sum, const float* __restrict__ gcp,
                                                        int ng, const Stats* __restrict__ st, float* __restrict__ dpred) {
    __shared__ double red[3][256];
    double a = 0, g1 = 0, g2 = 0;
    for (int k = threadIdx.x; k < nb_sum; k += 256) a += sum_partial[k];
    for (int k = threadIdx.x; k < ng; k += 256) { g1 += gcp[(size_t)k * 2]; g2 += gcp[(size_t)k * 2 + 1]; }
    red[0][threadIdx.x] = a; red[1][threadIdx.x] = g1; red[2][threadIdx.x] = g2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            for (int j = 0; j < 3; ++j) red[j][threadIdx.x] += red[j][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    const float drange = fmaxf(st->maxT, st->maxP) - fminf(st->minT, st->minP);
    const float ddr = (float)red[1][0] * 2.f * 0.01f * 0.01f * drange + (float)red[2][0] * 2.f * 0.03f * 0.03f * drange;
    if (st->maxP > st->maxT) dpred[st->argmaxP] += ddr;
    if (st->minP < st->minT) dpred[st->argminP] -= ddr;
    if (st->minP < 0.f) dpred[st->argminP] -= (float)red[0][0];
}

struct MsLayout {
    size_t stats, pf, pi, imgs, grads, maps, part, part2, gcp, means, gw, total;
    size_t img_off[MS_SCALES], grad_off[MS_SCALES], gcp_off[MS_SCALES];
    MsDims d;
    int nb_mm;
};
MsLayout ms_layout(int N, int H, int W, int C) {
    MsLayout l;
    l.d.H[0] = H; l.d.W[0] = W;
    for (int k = 1; k < MS_SCALES; ++k) { l.d.H[k] = (l.d.H[k - 1] + 1) / 2; l.d.W[k] = (l.d.W[k - 1] + 1) / 2; }
    l.nb_mm = 512;
    size_t o = 0;
    auto bump = [&](size_t bytes) { size_t r = o; o += (bytes + 255) & ~(size_t)255; return r; };
    l.stats = bump(sizeof(Stats));
    l.pf = bump((size_t)l.nb_mm * 4 * sizeof(float));
    l.pi = bump((size_t)l.nb_mm * 2 * sizeof(unsigned long long));
    size_t img = 0, grad = 0, gcp = 0;
    for (int k = 0; k < MS_SCALES; ++k) {
        const size_t n = (size_t)N * l.d.H[k] * l.d.W[k] * C;
        if (k > 0) { l.img_off[k] = img; img += 2 * n; } else l.img_off[k] = 0;      // x_k then q_k (k >= 1)
        l.grad_off[k] = grad; grad += n;
        l.gcp_off[k] = gcp;
        gcp += (size_t)N * C * cdiv(l.d.H[k] - KF + 1, TS) * cdiv(l.d.W[k] - KF + 1, TS);
    }
    l.imgs = bump(img * sizeof(float));
    l.grads = bump(grad * sizeof(float));
    l.maps = bump(3 * (size_t)N * C * (H - KF + 1) * (W - KF + 1) * sizeof(float));
    l.part = bump(std::max<size_t>((size_t)N * C * cdiv(H - KF + 1, TS) * cdiv(W - KF + 1, TS) * 2, 4096) * sizeof(float));
    l.part2 = bump((size_t)COMPACT_G * 2 * sizeof(float));
    l.gcp = bump(gcp * 2 * sizeof(float));
    l.means = bump((size_t)2 * MS_SCALES * N * C * sizeof(float));
    l.gw = bump((size_t)2 * MS_SCALES * N * C * sizeof(float));
    l.total = o;
    return l;
}
}  // namespace

size_t dssim_workspace_bytes(int N, int H, int W, int C) { return layout(N, H, W, C).total; }

void dssim_forward_backward(hipStream_t s, const float* y_true, const float* y_pred, float* dpred, int N, int H, int W, int C,
                            float weight, float* loss_out, int accumulate_loss, float* workspace, size_t workspace_bytes) {
    DL4DS_REQUIRE(H >= KF && W >= KF, "dssim needs grids of at least 11x11");
    Layout l = layout(N, H, W, C);
    DL4DS_REQUIRE(workspace_bytes >= l.total, "dssim workspace too small");
    char* base = reinterpret_cast<char*>(workspace);
    Stats* st = reinterpret_cast<Stats*>(base + l.stats);
    float* pf = reinterpret_cast<float*>(base + l.pf);
    unsigned long long* pi = reinterpret_cast<unsigned long long*>(base + l.pi);
    const int Ho = H - KF + 1, Wo = W - KF + 1;
    const size_t msz = (size_t)N * C * Ho * Wo;
    float* dmu = reinterpret_cast<float*>(base + l.maps);
    float* da = dmu + msz;
    float* db = da + msz;
    float* part = reinterpret_cast<float*>(base + l.part);
    const size_t n = (size_t)N * H * W * C;
    static const Gauss gk = make_gauss();
    ProfScope ps(s, "dssim", 0.0, 4.0 * (double)n * 8);
    const int nb = (int)std::min<size_t>(l.nb_mm, cdivz(n, 256));
    DL4DS_LAUNCH(minmax_kernel, dim3(nb), dim3(256), 0, s, y_true, y_pred, n, pf, pi);
    DL4DS_LAUNCH(minmax_finish_kernel, dim3(1), dim3(256), 0, s, pf, pi, nb, st);
    const int txo = cdiv(Wo, TS), tyo = cdiv(Ho, TS);
    const int nbf = N * C * txo * tyo;
    DL4DS_LAUNCH(ssim_fwd_kernel, dim3(nbf), dim3(256), 0, s, y_true, y_pred, H, W, C, Ho, Wo, txo, tyo, gk, st, dmu, da,
                       db, part);
    float* part2 = reinterpret_cast<float*>(base + l.part2);
    DL4DS_LAUNCH(compact_partials_kernel<3>, dim3(COMPACT_G), dim3(256), 0, s, part, nbf, part2);
    DL4DS_LAUNCH(sum3_kernel, dim3(1), dim3(256), 0, s, part2, COMPACT_G, st);
    const float inv_m = 1.f / (float)msz;
    const float coef = -0.5f * weight * inv_m;
    int nbb = 0;
    if (dpred) {
        const int txi = cdiv(W, TS), tyi = cdiv(H, TS);
        nbb = N * C * txi * tyi;
        DL4DS_LAUNCH(ssim_bwd_kernel, dim3(nbb), dim3(256), 0, s, y_true, y_pred, H, W, C, Ho, Wo, txi, tyi, gk, st, dmu, da,
                           db, coef, dpred, 1, part);
    }
    if (nbb) DL4DS_LAUNCH(compact_partials_kernel<1>, dim3(COMPACT_G), dim3(256), 0, s, part, nbb, part2);
    DL4DS_LAUNCH(dssim_finish_kernel, dim3(1), dim3(256), 0, s, part2, nbb ? COMPACT_G : 0, st, weight, inv_m, coef, dpred,
                       loss_out, accumulate_loss);
    HIP_CHECK(hipGetLastError());
}

size_t msdssim_workspace_bytes(int N, int H, int W, int C) { return ms_layout(N, H, W, C).total; }

void msdssim_forward_backward(hipStream_t s, const float* y_true, const float* y_pred, float* dpred, int N, int H, int W, int C,
                              float weight, float* loss_out, int accumulate_loss, float* workspace, size_t workspace_bytes) {
    DL4DS_REQUIRE(((H + 7) / 8) >= KF && ((W + 7) / 8) >= KF,
                  "msdssim needs grids of at least 81x81 (four scales, 11-tap filter on the coarsest)");
    const MsLayout l = ms_layout(N, H, W, C);
    DL4DS_REQUIRE(workspace_bytes >= l.total, "msdssim workspace too small");
    char* base = reinterpret_cast<char*>(workspace);
    Stats* st = reinterpret_cast<Stats*>(base + l.stats);
    float* pf = reinterpret_cast<float*>(base + l.pf);
    unsigned long long* pi = reinterpret_cast<unsigned long long*>(base + l.pi);
    float* imgs = reinterpret_cast<float*>(base + l.imgs);
    float* grads = reinterpret_cast<float*>(base + l.grads);
    float* maps = reinterpret_cast<float*>(base + l.maps);
    float* part = reinterpret_cast<float*>(base + l.part);
    float* gcp = reinterpret_cast<float*>(base + l.gcp);
    float* means = reinterpret_cast<float*>(base + l.means);       // [ssim: scales x NC][cs: scales x NC]
    float* gw = reinterpret_cast<float*>(base + l.gw);             // [gs: scales x NC][gc: scales x NC]
    const int NC = N * C;
    const size_t n0 = (size_t)N * H * W * C;
    static const Gauss gk = make_gauss();
    ProfScope ps(s, "msdssim", 0.0, 4.0 * (double)n0 * 12);
    const int nb = (int)std::min<size_t>(l.nb_mm, cdivz(n0, 256));
    DL4DS_LAUNCH(minmax_kernel, dim3(nb), dim3(256), 0, s, y_true, y_pred, n0, pf, pi);
    DL4DS_LAUNCH(minmax_finish_kernel, dim3(1), dim3(256), 0, s, pf, pi, nb, st);
    auto ew = [](size_t n) { return (int)std::max<size_t>(1, std::min<size_t>(cdivz(n, 256), 8192)); };
    const float* xk[MS_SCALES];
    const float* qk[MS_SCALES];
    xk[0] = y_true; qk[0] = y_pred;
    // ---- pyramids
    for (int k = 1; k < MS_SCALES; ++k) {
        const size_t n = (size_t)N * l.d.H[k] * l.d.W[k] * C;
        float* xd = imgs + l.img_off[k];
        float* qd = xd + n;
        DL4DS_LAUNCH(ms_pool_kernel, dim3(ew(n)), dim3(256), 0, s, xk[k - 1], xd, N, l.d.H[k - 1], l.d.W[k - 1], l.d.H[k],
                           l.d.W[k], C, st, k == 1 ? 1 : 0);
        DL4DS_LAUNCH(ms_pool_kernel, dim3(ew(n)), dim3(256), 0, s, qk[k - 1], qd, N, l.d.H[k - 1], l.d.W[k - 1], l.d.H[k],
                           l.d.W[k], C, st, k == 1 ? 2 : 0);
        xk[k] = xd; qk[k] = qd;
    }
    // ---- per-scale (ssim, cs) means
    for (int k = 0; k < MS_SCALES; ++k) {
        const int Ho = l.d.H[k] - KF + 1, Wo = l.d.W[k] - KF + 1;
        const int txo = cdiv(Wo, TS), tyo = cdiv(Ho, TS);
        DL4DS_LAUNCH(ms_ssim_kernel<0>, dim3(NC * txo * tyo), dim3(256), 0, s, xk[k], qk[k], k == 0 ? 1 : 0, l.d.H[k], l.d.W[k],
                           C, Ho, Wo, txo, tyo, gk, st, nullptr, nullptr, nullptr, nullptr, nullptr, part);
        DL4DS_LAUNCH(ms_means_kernel, dim3(NC), dim3(64), 0, s, part, txo * tyo, NC, 1.f / ((float)Ho * (float)Wo),
                           means + (size_t)k * NC, means + (size_t)(MS_SCALES + k) * NC);
    }
    DL4DS_LAUNCH(ms_combine_kernel, dim3(1), dim3(256), 0, s, means, means + (size_t)MS_SCALES * NC, NC, weight, gw,
                       gw + (size_t)MS_SCALES * NC, loss_out, accumulate_loss);
    HIP_CHECK(hipGetLastError());
    if (!dpred) return;
    // ---- per-scale gradients (coarse to fine), pooled scales folded into the finer one
    float* dmu = maps;
    size_t gcp_total = 0;
    for (int k = MS_SCALES - 1; k >= 0; --k) {
        const int Hk = l.d.H[k], Wk = l.d.W[k], Ho = Hk - KF + 1, Wo = Wk - KF + 1;
        const size_t msz = (size_t)NC * Ho * Wo;
        float* da = dmu + msz;
        float* db = da + msz;
        const int txo = cdiv(Wo, TS), tyo = cdiv(Ho, TS);
        DL4DS_LAUNCH(ms_ssim_kernel<1>, dim3(NC * txo * tyo), dim3(256), 0, s, xk[k], qk[k], k == 0 ? 1 : 0, Hk, Wk, C, Ho, Wo,
                           txo, tyo, gk, st, gw + (size_t)k * NC, gw + (size_t)(MS_SCALES + k) * NC, dmu, da, db,
                           gcp + l.gcp_off[k] * 2);
        gcp_total = std::max(gcp_total, l.gcp_off[k] + (size_t)NC * txo * tyo);
        const int txi = cdiv(Wk, TS), tyi = cdiv(Hk, TS);
        float* gk_img = grads + l.grad_off[k];
        DL4DS_LAUNCH(ms_bwd_kernel, dim3(NC * txi * tyi), dim3(256), 0, s, xk[k], qk[k], k == 0 ? 1 : 0, Hk, Wk, C, Ho, Wo, txi,
                           tyi, gk, st, dmu, da, db, gk_img);
        if (k < MS_SCALES - 1) {
            const size_t n = (size_t)N * Hk * Wk * C;
            DL4DS_LAUNCH(ms_unpool_add_kernel, dim3(ew(n)), dim3(256), 0, s, gk_img, grads + l.grad_off[k + 1], N, Hk, Wk,
                               l.d.H[k + 1], l.d.W[k + 1], C);
        }
    }
    const int nba = (int)std::min<size_t>(cdivz(n0, 256), 2048);
    DL4DS_LAUNCH(ms_apply_kernel, dim3(nba), dim3(256), 0, s, grads + l.grad_off[0], dpred, n0, 1, part);
    float* part2 = reinterpret_cast<float*>(base + l.part2);
    DL4DS_LAUNCH(compact_partials_kernel<2>, dim3(COMPACT_G), dim3(256), 0, s, gcp, (int)gcp_total, part2);
    DL4DS_LAUNCH(ms_finish_kernel, dim3(1), dim3(256), 0, s, part, nba, part2, COMPACT_G, st, dpred);
    HIP_CHECK(hipGetLastError());
}

// ============================================================================================ post-hoc image metrics
// dl4ds/metrics.py:166-185,196-262 (compute_metrics) without the plotting: per test pair PSNR (tf.image.psnr with the
// joint dynamic range), SSIM (tf.image.ssim, no min-shift here), MAE, RMSE and Pearson correlation over the grid; per grid
// point RMSE, mean bias and Pearson correlation over the pairs.  Everything is a streaming pass over the two arrays:
//   pair sums   : blocks of one pair reduce (|d|, d^2, x, y, xy, x^2, y^2) -> fixed-order partials -> one thread per pair;
//   grid points : one thread per (pixel, channel) walks the pairs (coalesced across pixels);
//   SSIM        : the tile kernel of the multi-scale loss at scale 0 (separable Gaussian moments) + its per-plane means.
namespace {
constexpr int MET_CHUNKS = 64, MET_K = 7;

__global__ void __launch_bounds__(256) met_pair_partial_kernel(const float* __restrict__ t, const float* __restrict__ p, size_t per,
                                                               float* __restrict__ partial) {
    __shared__ double red[MET_K][256];
    const int n = blockIdx.y;
    const size_t chunk = (per + gridDim.x - 1) / gridDim.x;
    const size_t e0 = (size_t)blockIdx.x * chunk, e1 = min(e0 + chunk, per);
    const float* a = t + (size_t)n * per;
    const float* b = p + (size_t)n * per;
    double s[MET_K] = {0, 0, 0, 0, 0, 0, 0};
    for (size_t e = e0 + threadIdx.x; e < e1; e += 256) {
        const double x = a[e], y = b[e], d = y - x;
        s[0] += fabs(d); s[1] += d * d; s[2] += x; s[3] += y; s[4] += x * y; s[5] += x * x; s[6] += y * y;
    }
    for (int k = 0; k < MET_K; ++k) red[k][threadIdx.x] = s[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            for (int k = 0; k < MET_K; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x < MET_K) partial[((size_t)n * gridDim.x + blockIdx.x) * MET_K + threadIdx.x] = (float)red[threadIdx.x][0];
}
// out[n] = (mae, mse, pearson over the grid)
__global__ void met_pair_finish_kernel(const float* __restrict__ partial, int nchunks, int N, double per, float* __restrict__ out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double s[MET_K] = {0, 0, 0, 0, 0, 0, 0};
    for (int c = 0; c < nchunks; ++c)
        for (int k = 0; k < MET_K; ++k) s[k] += (double)partial[((size_t)n * nchunks + c) * MET_K + k];
    const double mx = s[2] / per, my = s[3] / per;
    const double cov = s[4] / per - mx * my, vx = s[5] / per - mx * mx, vy = s[6] / per - my * my;
    out[n * 3 + 0] = (float)(s[0] / per);
    out[n * 3 + 1] = (float)(s[1] / per);
    out[n * 3 + 2] = (float)(cov / sqrt(fmax(vx * vy, 1e-300)));
}
// maps[0][e] = sqrt(mean_n d^2), maps[1][e] = mean_n d, maps[2][e] = pearson over the pairs
__global__ void met_grid_kernel(const float* __restrict__ t, const float* __restrict__ p, int N, size_t per, float* __restrict__ maps) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < per; e += (size_t)gridDim.x * blockDim.x) {
        double sd = 0, sdd = 0, sx = 0, sy = 0, sxy = 0, sxx = 0, syy = 0;
        for (int n = 0; n < N; ++n) {
            const double x = t[(size_t)n * per + e], y = p[(size_t)n * per + e], d = y - x;
            sd += d; sdd += d * d; sx += x; sy += y; sxy += x * y; sxx += x * x; syy += y * y;
        }
        const double nn = (double)N, mx = sx / nn, my = sy / nn;
        const double cov = sxy / nn - mx * my, vx = sxx / nn - mx * mx, vy = syy / nn - my * my;
        maps[e] = (float)sqrt(sdd / nn);
        maps[per + e] = (float)(sd / nn);
        maps[2 * per + e] = (vx > 0 && vy > 0) ? (float)(cov / sqrt(vx * vy)) : NAN;
    }
}
__global__ void met_assemble_kernel(const float* __restrict__ pair3, const float* __restrict__ means, int N, int C, int has_ssim,
                                    const Stats* __restrict__ st, float* __restrict__ out, float* __restrict__ range) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n == 0) {
        range[0] = fminf(st->minT, st->minP);
        range[1] = fmaxf(st->maxT, st->maxP);
    }
    if (n >= N) return;
    float sm = 0.f;
    for (int c = 0; c < C; ++c) sm += has_ssim ? means[n * C + c] : NAN;
    out[n * 4 + 0] = pair3[n * 3 + 0];
    out[n * 4 + 1] = pair3[n * 3 + 1];
    out[n * 4 + 2] = pair3[n * 3 + 2];
    out[n * 4 + 3] = sm / (float)C;
}
}  // namespace

size_t metrics_workspace_bytes(int N, int H, int W, int C) {
    const size_t per = (size_t)H * W * C;
    const int Ho = H - KF + 1, Wo = W - KF + 1;
    const size_t tiles = (Ho > 0 && Wo > 0) ? (size_t)N * C * cdiv(Ho, TS) * cdiv(Wo, TS) : 0;
    return sizeof(Stats) + 256 + (size_t)512 * (4 * sizeof(float) + 2 * sizeof(unsigned long long)) + 256 +
           ((size_t)N * MET_CHUNKS * MET_K + (size_t)N * 3 + 3 * per + tiles * 2 + (size_t)2 * N * C + 64) * sizeof(float);
}

// pair_out: [N][4] = (mae, mse, pearson, ssim)   grid_out: [3][H*W*C] = (rmse, mean bias, pearson)   stats_out: Stats
void image_metrics(hipStream_t s, const float* y_true, const float* y_pred, int N, int H, int W, int C, float* pair_out_dev,
                   float* grid_out_dev, float* range_out_dev, float* workspace, size_t workspace_bytes) {
    DL4DS_REQUIRE(workspace_bytes >= metrics_workspace_bytes(N, H, W, C), "metrics workspace too small");
    const size_t per = (size_t)H * W * C, n0 = per * N;
    char* base = reinterpret_cast<char*>(workspace);
    Stats* st = reinterpret_cast<Stats*>(base);
    float* pf = reinterpret_cast<float*>(base + 256);
    unsigned long long* pi = reinterpret_cast<unsigned long long*>(base + 256 + 512 * 4 * sizeof(float));
    float* f = reinterpret_cast<float*>(base + 256 + 512 * (4 * sizeof(float) + 2 * sizeof(unsigned long long)) + 256);
    float* partial = f;                                   f += (size_t)N * MET_CHUNKS * MET_K;
    float* pair3 = f;                                     f += (size_t)N * 3;
    float* tile_part = f;
    static const Gauss gk = make_gauss();
    ProfScope ps(s, "image_metrics", 0.0, 4.0 * (double)n0 * 6);
    const int nb = (int)std::min<size_t>(512, cdivz(n0, 256));
    DL4DS_LAUNCH(minmax_kernel, dim3(nb), dim3(256), 0, s, y_true, y_pred, n0, pf, pi);
    DL4DS_LAUNCH(minmax_finish_kernel, dim3(1), dim3(256), 0, s, pf, pi, nb, st);
    DL4DS_LAUNCH(met_pair_partial_kernel, dim3(MET_CHUNKS, N), dim3(256), 0, s, y_true, y_pred, per, partial);
    DL4DS_LAUNCH(met_pair_finish_kernel, dim3(cdiv(N, 64)), dim3(64), 0, s, partial, MET_CHUNKS, N, (double)per, pair3);
    DL4DS_LAUNCH(met_grid_kernel, dim3((unsigned)std::min<size_t>(cdivz(per, 256), 4096)), dim3(256), 0, s, y_true, y_pred, N,
                       per, grid_out_dev);
    // SSIM per pair = mean over channels of the per-plane mean SSIM (needs an 11x11 window)
    const int Ho = H - KF + 1, Wo = W - KF + 1;
    float* means = tile_part;           // set below when the window fits
    const int NC = N * C;
    if (Ho > 0 && Wo > 0) {
        const int txo = cdiv(Wo, TS), tyo = cdiv(Ho, TS);
        means = tile_part + (size_t)NC * txo * tyo * 2;
        DL4DS_LAUNCH(ms_ssim_kernel<0>, dim3(NC * txo * tyo), dim3(256), 0, s, y_true, y_pred, 0, H, W, C, Ho, Wo, txo, tyo, gk, st,
                           nullptr, nullptr, nullptr, nullptr, nullptr, tile_part);
        DL4DS_LAUNCH(ms_means_kernel, dim3(NC), dim3(64), 0, s, tile_part, txo * tyo, NC, 1.f / ((float)Ho * (float)Wo), means,
                           means + NC);
    }
    HIP_CHECK(hipGetLastError());
    // assemble [N][4] and the range on the device (tiny)
    DL4DS_LAUNCH(met_assemble_kernel, dim3(cdiv(N, 64)), dim3(64), 0, s, pair3, means, N, C, (Ho > 0 && Wo > 0) ? 1 : 0, st, pair_out_dev,
                       range_out_dev);
    HIP_CHECK(hipGetLastError());
}
