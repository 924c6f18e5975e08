// dl4ds_amd -- on-device batch preparation ("next" row f1): the step immediately BEFORE the train step.
//
// Replaces the per-sample Python/cv2 loop of dl4ds/dataloader.py:11-360 (create_pair_hr_lr / create_batch_hr_lr) and
// dl4ds/utils.py:251-401 (crop_array / resize_array) for the default path the trainers take: 'inter_area'
// interpolation (cv2.INTER_AREA at an integer ratio == block mean; at an integer up-scaling ratio == pixel
// replication), no externally supplied LR array.  The whole dataset (HR fields, optional predictor fields, optional
// static variables) lives in HBM; a batch is produced by two gather kernels from (sample index, crop corner) lists
// drawn on the host with the same RNG calls as the numpy port, so both paths yield the same batches:
//   post-upsampling :  lr[b,t,y,x,:] = [ blockmean_s(hr) | blockmean_s(predictors) | blockmean_s(static) ]   (LR grid)
//   pre-upsampling  :  lr[b,t,Y,X,:] = [ replicate(blockmean_s(hr)) | replicate(blockmean_s(predictors)) | static ]
//   hr[b,t,Y,X,:]   =  crop of the HR field,   static_hr[b,Y,X,:] = crop of the static variables
// (static variables are appended to lr only for spatial models, dataloader.py:261-289 -> `static_in_lr`).
// Pure HBM streaming: every HR pixel of the crop is read once for the block means and once for the HR copy.
#include "ops.h"
#include "prof.h"
#include <algorithm>

namespace {

struct PrepParams {
    const float* hr;        // [N][H][W][C]
    const float* pred;      // [N][H][W][P] or null
    const float* stat;      // [H][W][S] or null
    const int* idx;         // [B] first frame of each sample
    const int* cy;          // [B] crop corner in HR pixels
    const int* cx;
    float* out_lr;          // post: [B][T][ps/s][ps/s][CL]   pin: [B][T][ps][ps][CL]
    float* out_hr;          // [B][T][ps][ps][C]
    float* out_stat;        // [B][ps][ps][S] or null
    int H, W, C, P, S, T, B;
    int scale, ps;          // ps = HR patch size (== H == W when not cropping; H, W may differ -> psy, psx)
    int psy, psx;
    int pin;                // 1: LR is re-expanded to the HR grid
    int static_in_lr;       // append (block-mean / raw) static variables to lr
};

// mean over the scale x scale block whose top-left PIXEL is (py, px) of channel c of a [H][W][Cn] image
__device__ __forceinline__ float block_mean(const float* __restrict__ img, int W, int Cn, int c, int py, int px, int s) {
    float acc = 0.f;
    const float* p = img + ((size_t)py * W + (size_t)px) * Cn + c;
    for (int dy = 0; dy < s; ++dy)
        for (int dx = 0; dx < s; ++dx) acc += p[((size_t)dy * W + dx) * Cn];
    return acc / (float)(s * s);
}

__global__ void prep_lr_kernel(const PrepParams a) {
    const int CL = a.C + a.P + (a.static_in_lr ? a.S : 0);
    const int oy_n = a.pin ? a.psy : a.psy / a.scale, ox_n = a.pin ? a.psx : a.psx / a.scale;
    const size_t total = (size_t)a.B * a.T * oy_n * ox_n * CL;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % CL);
        size_t r = e / CL;
        const int ox = (int)(r % ox_n); r /= ox_n;
        const int oy = (int)(r % oy_n); r /= oy_n;
        const int t = (int)(r % a.T);
        const int b = (int)(r / a.T);
        // absolute HR position of this output pixel (pin) or of its block's corner (post)
        const int Y = a.pin ? a.cy[b] + oy : a.cy[b] + oy * a.scale;
        const int X = a.pin ? a.cx[b] + ox : a.cx[b] + ox * a.scale;
        // 'pin': the WHOLE field is coarsened and re-expanded before the crop (dataloader.py:102-106), so the block is
        // the one of the field's own LR grid that contains the pixel; post-upsampling: the crop is taken first (at any
        // pixel) and the PATCH is coarsened (dataloader.py:201-205), so blocks are aligned with the crop corner
        const int by = a.pin ? (Y / a.scale) * a.scale : Y, bx = a.pin ? (X / a.scale) * a.scale : X;
        const size_t frame = (size_t)(a.idx[b] + t) * a.H * a.W;
        float v;
        if (c < a.C) {
            v = block_mean(a.hr + frame * a.C, a.W, a.C, c, by, bx, a.scale);
        } else if (c < a.C + a.P) {
            v = block_mean(a.pred + frame * a.P, a.W, a.P, c - a.C, by, bx, a.scale);
        } else if (a.pin) {
            v = a.stat[((size_t)Y * a.W + X) * a.S + (c - a.C - a.P)];
        } else {
            v = block_mean(a.stat, a.W, a.S, c - a.C - a.P, by, bx, a.scale);
        }
        a.out_lr[e] = v;
    }
}

__global__ void prep_hr_kernel(const PrepParams a) {
    const size_t n_hr = (size_t)a.B * a.T * a.psy * a.psx * a.C;
    const size_t n_st = a.out_stat ? (size_t)a.B * a.psy * a.psx * a.S : 0;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_hr + n_st; e += (size_t)gridDim.x * blockDim.x) {
        if (e < n_hr) {
            const int c = (int)(e % a.C);
            size_t r = e / a.C;
            const int x = (int)(r % a.psx); r /= a.psx;
            const int y = (int)(r % a.psy); r /= a.psy;
            const int t = (int)(r % a.T);
            const int b = (int)(r / a.T);
            a.out_hr[e] = a.hr[(((size_t)(a.idx[b] + t) * a.H + a.cy[b] + y) * a.W + a.cx[b] + x) * a.C + c];
        } else {
            const size_t q = e - n_hr;
            const int c = (int)(q % a.S);
            size_t r = q / a.S;
            const int x = (int)(r % a.psx); r /= a.psx;
            const int y = (int)(r % a.psy);
            const int b = (int)(r / a.psy);
            a.out_stat[q] = a.stat[((size_t)(a.cy[b] + y) * a.W + a.cx[b] + x) * a.S + c];
        }
    }
}

// ---- every other interpolation of resize_array (utils.py:330-401: cv2 nearest / bilinear / bicubic / Lanczos-4, and
// inter_area again): cv2.resize is separable, so one axis of it is a table of K (source index, weight) taps per output
// row / column (built once on the host by the cv2 restatement, resident in HBM).  A "gather" pass evaluates
//     out[b,t,oy,ox,c] = sum_ky sum_kx wy[ry,ky] wx[rx,kx] src_c[y0 + iy[ry,ky], x0 + ix[rx,kx]]
// for up to three channel groups with their own source, table and origin rule (see batch_prepare_taps below).
struct TapTable {
    const int* iy; const float* wy; int ky;
    const int* ix; const float* wx; int kx;
};
struct TapGroup {
    const float* src;
    int Cn;                 // channels of the source image
    int frames;             // 0: dataset frame idx[b]+t   1: batch-local frame b*T+t   2: one static image
    int sh, sw;             // source image size
    int raw;                // 1: copy the source pixel at (cy+oy, cx+ox), no taps
    int origin_from_crop;   // 1: tap indices are relative to the crop corner (the PATCH was resized)
    int row_div;            // >0: table row = oy + cy/row_div (the whole FIELD was resized, then cropped)
    TapTable tab;
};
struct TapParams {
    TapGroup g[3];
    int cend[3];            // exclusive channel end of each group in the output
    int ng;
    const int* idx; const int* cy; const int* cx;   // cy/cx null: no crop (origin 0)
    float* out;
    int oy_n, ox_n, CL, T, B;
};

__global__ void prep_taps_kernel(const TapParams a) {
    const size_t total = (size_t)a.B * a.T * a.oy_n * a.ox_n * a.CL;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % a.CL);
        size_t r = e / a.CL;
        const int ox = (int)(r % a.ox_n); r /= a.ox_n;
        const int oy = (int)(r % a.oy_n); r /= a.oy_n;
        const int t = (int)(r % a.T);
        const int b = (int)(r / a.T);
        int gi = 0;
        while (gi + 1 < a.ng && c >= a.cend[gi]) ++gi;
        const TapGroup& g = a.g[gi];
        const int cc = c - (gi ? a.cend[gi - 1] : 0);
        const int cy = a.cy ? a.cy[b] : 0, cx = a.cx ? a.cx[b] : 0;
        const size_t frame = g.frames == 0 ? (size_t)(a.idx[b] + t) : g.frames == 1 ? (size_t)b * a.T + t : 0;
        const float* img = g.src + frame * g.sh * g.sw * g.Cn + cc;
        float v;
        if (g.raw) {
            // copy of the source pixel at the crop corner (+ output position); row_div > 0: the source lives on a grid row_div times
            // coarser than the corner's (an external LR array / LR-grid predictors cropped at corner / scale, dataloader.py:166-200)
            const int sy = g.row_div ? cy / g.row_div : cy, sx = g.row_div ? cx / g.row_div : cx;
            v = img[((size_t)(sy + oy) * g.sw + sx + ox) * g.Cn];
        } else {
            const int ry = g.row_div ? oy + cy / g.row_div : oy, rx = g.row_div ? ox + cx / g.row_div : ox;
            const int y0 = g.origin_from_crop ? cy : 0, x0 = g.origin_from_crop ? cx : 0;
            const int* iy = g.tab.iy + (size_t)ry * g.tab.ky;
            const float* wy = g.tab.wy + (size_t)ry * g.tab.ky;
            const int* ix = g.tab.ix + (size_t)rx * g.tab.kx;
            const float* wx = g.tab.wx + (size_t)rx * g.tab.kx;
            v = 0.f;
            for (int ky = 0; ky < g.tab.ky; ++ky) {
                const float* row = img + (size_t)(y0 + iy[ky]) * g.sw * g.Cn;
                float acc = 0.f;
                for (int kx = 0; kx < g.tab.kx; ++kx) acc += wx[kx] * row[(size_t)(x0 + ix[kx]) * g.Cn];
                v += wy[ky] * acc;
            }
        }
        a.out[e] = v;
    }
}

void launch_taps(hipStream_t s, const TapParams& a, const char* name, double src_bytes) {
    const size_t n = (size_t)a.B * a.T * a.oy_n * a.ox_n * a.CL;
    ProfScope ps(s, name, 0.0, src_bytes + 4.0 * n);
    const int blocks = (int)std::max<size_t>(1, std::min<size_t>(cdivz(n, 256), 16384));
    DL4DS_LAUNCH(prep_taps_kernel, dim3(blocks), dim3(256), 0, s, a);
    HIP_CHECK(hipGetLastError());
}

}  // namespace

// Batch preparation for any separable interpolation.  Tables (device, built by the caller from cv2's coefficients):
//   dn_patch : [psy/scale][ky] / [psx/scale][kx]  resize of a psy x psx PATCH to the LR grid (indices relative to the patch)
//   dn_field : [H/scale] / [W/scale]               resize of the whole field to the LR grid
//   up_field : [H] / [W]                           resize of the LR field back to H x W ('pin')
// post-upsampling (dataloader.py:141-205): the HR crop is resized (dn_patch); predictors are resized as whole fields and
// cropped on the LR grid (dn_field, corner / scale); static variables are cropped, then resized (dn_patch).
// 'pin' (dataloader.py:94-112): HR and predictor fields are resized to the LR grid (dn_field, into `scratch`
// [B][T][H/scale][W/scale][C+P]) and back (up_field), then cropped; static variables are cropped only.
void batch_prepare_taps(hipStream_t s, const float* hr, const float* pred, const float* stat, const int* idx, const int* cy,
                        const int* cx, float* out_lr, float* out_hr, float* out_stat, float* scratch, int H, int W, int C,
                        int P, int S, int T, int B, int scale, int psy, int psx, int pin, int static_in_lr,
                        const TapAxis* dn_patch, const TapAxis* dn_field, const TapAxis* up_field) {
    DL4DS_REQUIRE(B > 0 && T > 0 && C > 0 && scale >= 1, "batch_prepare_taps: bad sizes");
    DL4DS_REQUIRE(psy <= H && psx <= W, "batch_prepare_taps: patch larger than the field");
    DL4DS_REQUIRE((P == 0) == (pred == nullptr) && (S == 0) == (stat == nullptr), "batch_prepare_taps: predictor/static pointers");
    auto table = [](const TapAxis* t) {
        TapTable r;
        r.iy = t[0].idx; r.wy = t[0].wt; r.ky = t[0].k; r.ix = t[1].idx; r.wx = t[1].wt; r.kx = t[1].k;
        return r;
    };
    auto ok = [](const TapAxis* t) { return t && t[0].idx && t[0].wt && t[0].k > 0 && t[1].idx && t[1].wt && t[1].k > 0; };
    const int hl = H / scale, wl = W / scale;
    const bool with_stat = static_in_lr && S;
    TapParams a{};
    a.idx = idx; a.T = T; a.B = B;
    if (pin) {
        DL4DS_REQUIRE(ok(dn_field) && ok(up_field) && scratch, "batch_prepare_taps: 'pin' needs dn_field, up_field and scratch");
        // pass 1: whole fields -> LR grid
        a.cy = a.cx = nullptr;
        a.out = scratch; a.oy_n = hl; a.ox_n = wl; a.CL = C + P; a.ng = 0;
        auto field = [&](const float* src, int Cn) {
            TapGroup& g = a.g[a.ng];
            g = TapGroup{};
            g.src = src; g.Cn = Cn; g.frames = 0; g.sh = H; g.sw = W; g.tab = table(dn_field);
            a.cend[a.ng] = (a.ng ? a.cend[a.ng - 1] : 0) + Cn;
            ++a.ng;
        };
        field(hr, C);
        if (P) field(pred, P);
        launch_taps(s, a, "batch_prepare_taps_down", 4.0 * B * T * (double)H * W * (C + P));
        // pass 2: LR fields -> HR grid, cropped; static variables copied
        a.cy = cy; a.cx = cx;
        a.out = out_lr; a.oy_n = psy; a.ox_n = psx; a.CL = C + P + (with_stat ? S : 0); a.ng = 1;
        a.g[0] = TapGroup{};
        a.g[0].src = scratch; a.g[0].Cn = C + P; a.g[0].frames = 1; a.g[0].sh = hl; a.g[0].sw = wl; a.g[0].row_div = 1;
        a.g[0].tab = table(up_field);
        a.cend[0] = C + P;
        if (with_stat) {
            a.g[1] = TapGroup{};
            a.g[1].src = stat; a.g[1].Cn = S; a.g[1].frames = 2; a.g[1].sh = H; a.g[1].sw = W; a.g[1].raw = 1;
            a.cend[1] = C + P + S;
            a.ng = 2;
        }
        launch_taps(s, a, "batch_prepare_taps_up", 4.0 * B * T * (double)hl * wl * (C + P));
    } else {
        DL4DS_REQUIRE(psy % scale == 0 && psx % scale == 0, "batch_prepare_taps: patch size must be divisible by scale");
        DL4DS_REQUIRE(ok(dn_patch) && (!P || ok(dn_field)), "batch_prepare_taps: tables missing");
        a.cy = cy; a.cx = cx;
        a.out = out_lr; a.oy_n = psy / scale; a.ox_n = psx / scale; a.CL = C + P + (with_stat ? S : 0); a.ng = 0;
        int cend = 0;
        auto add = [&](const float* src, int Cn, int frames, bool whole_field) {
            TapGroup& g = a.g[a.ng];
            g = TapGroup{};
            g.src = src; g.Cn = Cn; g.frames = frames; g.sh = H; g.sw = W;
            if (whole_field) { g.row_div = scale; g.tab = table(dn_field); }
            else { g.origin_from_crop = 1; g.tab = table(dn_patch); }
            cend += Cn;
            a.cend[a.ng++] = cend;
        };
        add(hr, C, 0, false);
        if (P) add(pred, P, 0, true);
        if (with_stat) add(stat, S, 2, false);
        launch_taps(s, a, "batch_prepare_taps_down", 4.0 * B * T * (double)psy * psx * (C + P));
    }
    // the HR crop and the static-variable crop are the copies of the default path
    PrepParams h{};
    h.hr = hr; h.stat = stat; h.idx = idx; h.cy = cy; h.cx = cx; h.out_hr = out_hr; h.out_stat = S ? out_stat : nullptr;
    h.H = H; h.W = W; h.C = C; h.S = S; h.T = T; h.B = B; h.psy = psy; h.psx = psx;
    const size_t n_hr = (size_t)B * T * psy * psx * C + (S ? (size_t)B * psy * psx * S : 0);
    ProfScope ps(s, "batch_prepare_hr", 0.0, 8.0 * n_hr);
    const int blocks = (int)std::max<size_t>(1, std::min<size_t>(cdivz(n_hr, 256), 16384));
    DL4DS_LAUNCH(prep_hr_kernel, dim3(blocks), dim3(256), 0, s, h);
    HIP_CHECK(hipGetLastError());
}

// The gather pass itself as a primitive (round 5): up to three channel groups, each with its own source array / grid / frame rule and
// either a raw crop or a separable tap table.  The host composes create_pair_hr_lr's remaining cases from it (an external LR array,
// predictors already on the LR grid, fields whose size `scale` does not divide -- dl4ds_amd/dataloader.py: DeviceDataGenerator).
void batch_gather(hipStream_t s, const GatherGroup* groups, int n_groups, const int* idx, const int* cy, const int* cx, float* out,
                  int out_h, int out_w, int T, int B) {
    DL4DS_REQUIRE(n_groups >= 1 && n_groups <= 3 && B > 0 && T > 0 && out_h > 0 && out_w > 0 && out, "batch_gather: bad sizes");
    TapParams a{};
    a.idx = idx; a.cy = cy; a.cx = cx; a.out = out; a.oy_n = out_h; a.ox_n = out_w; a.T = T; a.B = B; a.ng = n_groups;
    int cend = 0;
    double src_bytes = 0.0;
    for (int i = 0; i < n_groups; ++i) {
        const GatherGroup& q = groups[i];
        DL4DS_REQUIRE(q.src && q.channels > 0 && q.src_h > 0 && q.src_w > 0 && q.frames >= 0 && q.frames <= 2, "batch_gather: bad group");
        DL4DS_REQUIRE(q.frames != 0 || idx, "batch_gather: dataset-indexed group without an index list");
        DL4DS_REQUIRE(q.raw || (q.taps[0].idx && q.taps[0].wt && q.taps[0].k > 0 && q.taps[1].idx && q.taps[1].wt && q.taps[1].k > 0),
                      "batch_gather: tap tables missing");
        TapGroup& g = a.g[i];
        g = TapGroup{};
        g.src = q.src; g.Cn = q.channels; g.frames = q.frames; g.sh = q.src_h; g.sw = q.src_w; g.raw = q.raw;
        g.origin_from_crop = q.origin_from_crop; g.row_div = q.row_div;
        if (!q.raw) { g.tab.iy = q.taps[0].idx; g.tab.wy = q.taps[0].wt; g.tab.ky = q.taps[0].k; g.tab.ix = q.taps[1].idx; g.tab.wx = q.taps[1].wt; g.tab.kx = q.taps[1].k; }
        cend += q.channels;
        a.cend[i] = cend;
        src_bytes += 4.0 * B * T * (double)out_h * out_w * q.channels * (q.raw ? 1 : q.taps[0].k * q.taps[1].k);
    }
    a.CL = cend;
    launch_taps(s, a, "batch_gather", src_bytes);
}

void batch_prepare(hipStream_t s, const float* hr, const float* pred, const float* stat, const int* idx, const int* cy,
                   const int* cx, float* out_lr, float* out_hr, float* out_stat, int H, int W, int C, int P, int S, int T,
                   int B, int scale, int psy, int psx, int pin, int static_in_lr) {
    DL4DS_REQUIRE(B > 0 && T > 0 && C > 0 && scale >= 1, "batch_prepare: bad sizes");
    DL4DS_REQUIRE(H % scale == 0 && W % scale == 0, "batch_prepare: field size must be divisible by scale (block mean)");
    DL4DS_REQUIRE(psy <= H && psx <= W, "batch_prepare: patch larger than the field");
    DL4DS_REQUIRE(pin || (psy % scale == 0 && psx % scale == 0), "batch_prepare: patch size must be divisible by scale");
    DL4DS_REQUIRE((P == 0) == (pred == nullptr) && (S == 0) == (stat == nullptr), "batch_prepare: predictor/static pointers");
    PrepParams a;
    a.hr = hr; a.pred = pred; a.stat = stat; a.idx = idx; a.cy = cy; a.cx = cx;
    a.out_lr = out_lr; a.out_hr = out_hr; a.out_stat = S ? out_stat : nullptr;
    a.H = H; a.W = W; a.C = C; a.P = P; a.S = S; a.T = T; a.B = B; a.scale = scale; a.ps = psy; a.psy = psy; a.psx = psx;
    a.pin = pin; a.static_in_lr = (static_in_lr && S) ? 1 : 0;
    const int CL = C + P + (a.static_in_lr ? S : 0);
    const size_t n_lr = (size_t)B * T * (pin ? psy : psy / scale) * (pin ? psx : psx / scale) * CL;
    const size_t n_hr = (size_t)B * T * psy * psx * C + (S ? (size_t)B * psy * psx * S : 0);
    const double crop_bytes = 4.0 * B * T * (double)psy * psx * (C + P);
    {
        ProfScope ps(s, "batch_prepare_lr", 0.0, crop_bytes + 4.0 * n_lr);
        const int blocks = (int)std::max<size_t>(1, std::min<size_t>(cdivz(n_lr, 256), 16384));
        DL4DS_LAUNCH(prep_lr_kernel, dim3(blocks), dim3(256), 0, s, a);
        HIP_CHECK(hipGetLastError());
    }
    {
        ProfScope ps(s, "batch_prepare_hr", 0.0, 8.0 * n_hr);
        const int blocks = (int)std::max<size_t>(1, std::min<size_t>(cdivz(n_hr, 256), 16384));
        DL4DS_LAUNCH(prep_hr_kernel, dim3(blocks), dim3(256), 0, s, a);
        HIP_CHECK(hipGetLastError());
    }
}
