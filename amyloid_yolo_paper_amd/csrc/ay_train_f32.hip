// Training-step kernels of the fp32 (reference-precision, NCHW) path:
//   train-mode BatchNorm forward/backward fused with LeakyReLU (models.py:43-45, PyTorch momentum semantics),
//   convolution dgrad / wgrad / bias grad (autograd of models.py:33-40), shortcut / route / upsample plumbing
//   (models.py:86-96,244-248), YOLO target assignment + loss + head gradient (utils/utils.py:276-330, models.py:174-222),
//   Adam on a flat buffer (train.py:81,118).
// These are the correctness-first versions (plain fp32 FMA, one pass per tensor); the bf16 MFMA forward reuses
// ay_conv_bf16.hip.  Everything is deterministic except the float atomics of the loss sums (order-dependent last bits).
#include "ay_common.h"

namespace ay {

// ------------------------------------------------------------------------------------------ reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float block_sum(float v, float* sm) {  // 256 threads, result on every thread
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}

__device__ __forceinline__ double block_sum_d(double v, double* sm) {  // fp64 accumulation, as ATen's CPU BatchNorm does
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}

// ------------------------------------------------------------------------------------------ BatchNorm (train)
// one workgroup per channel; two-pass statistics (mean, then centred second moment), then normalise + affine + leaky.
__global__ void __launch_bounds__(256) bn_train_fwd_kernel(const float* __restrict__ z, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* running_mean,
                                                           float* running_var, float momentum, float eps, int leaky,
                                                           float* __restrict__ y, float* __restrict__ save_mean,
                                                           float* __restrict__ save_invstd, int B, int C, int HW) {
    __shared__ double sm[4];
    const int c = blockIdx.x;
    const long long n = (long long)B * HW;
    double s = 0.0;
    for (long long i = threadIdx.x; i < n; i += 256) s += (double)z[((i / HW) * C + c) * (long long)HW + i % HW];
    const double mean_d = block_sum_d(s, sm) / (double)n;
    double q = 0.0;
    for (long long i = threadIdx.x; i < n; i += 256) {
        const double d = (double)z[((i / HW) * C + c) * (long long)HW + i % HW] - mean_d;
        q += d * d;
    }
    const double var_d = block_sum_d(q, sm) / (double)n;  // biased, used for normalisation
    const float mean = (float)mean_d, var = (float)var_d;
    const float invstd = (float)(1.0 / sqrt(var_d + (double)eps));
    const float g = gamma[c], bt = beta[c];
    for (long long i = threadIdx.x; i < n; i += 256) {
        const long long o = ((i / HW) * C + c) * (long long)HW + i % HW;
        float v = (z[o] - mean) * invstd * g + bt;
        if (leaky) v = v > 0.f ? v : 0.1f * v;
        y[o] = v;
    }
    if (threadIdx.x == 0) {
        save_mean[c] = mean;
        save_invstd[c] = invstd;
        const float unbiased = n > 1 ? (float)(var_d * (double)n / (double)(n - 1)) : var;
        running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.0f - momentum) * running_var[c] + momentum * unbiased;
    }
}

// dy: gradient wrt the block output y = leaky(bn(z)); produces dz, dgamma, dbeta.
__global__ void __launch_bounds__(256) bn_train_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                           const float* __restrict__ z, const float* __restrict__ gamma,
                                                           const float* __restrict__ save_mean,
                                                           const float* __restrict__ save_invstd, int leaky,
                                                           float* __restrict__ dz, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int B, int C, int HW) {
    __shared__ double sm[4];
    const int c = blockIdx.x;
    const long long n = (long long)B * HW;
    const float mean = save_mean[c], invstd = save_invstd[c];
    double sb = 0.0, sg = 0.0;
    for (long long i = threadIdx.x; i < n; i += 256) {
        const long long o = ((i / HW) * C + c) * (long long)HW + i % HW;
        float d = dy[o];
        if (leaky && !(y[o] > 0.f)) d *= 0.1f;  // sign(y) == sign(pre-activation); torch: slope for x <= 0
        sb += (double)d;
        sg += (double)(d * ((z[o] - mean) * invstd));
    }
    const float db = (float)block_sum_d(sb, sm);
    const float dg = (float)block_sum_d(sg, sm);
    const float k = gamma[c] * invstd / (float)n;
    for (long long i = threadIdx.x; i < n; i += 256) {
        const long long o = ((i / HW) * C + c) * (long long)HW + i % HW;
        float d = dy[o];
        if (leaky && !(y[o] > 0.f)) d *= 0.1f;
        const float xh = (z[o] - mean) * invstd;
        dz[o] = k * ((float)n * d - db - xh * dg);
    }
    if (threadIdx.x == 0) {
        dgamma[c] = dg;
        dbeta[c] = db;
    }
}

// bias-only (linear head) backward: db[c] = sum dz
// db[c] (+)= sum over batch and pixels of dz[b][c][:]; grid (channels, chunks): a chunk is a contiguous range of one channel's
// B*HW values, summed in fixed order and added with ONE atomic (the db entries are zeroed first unless `accumulate`); with one
// workgroup per channel the 24-channel head gradients of a 1024^2 batch took 0.36 ms each
__global__ void __launch_bounds__(256) bias_grad_kernel(const float* __restrict__ dz, float* __restrict__ db, int B, int C, int HW) {
    __shared__ float sm[4];
    const int c = blockIdx.x;
    const long long n = (long long)B * HW;
    const long long per = (n + gridDim.y - 1) / gridDim.y;
    const long long lo = per * blockIdx.y, hi = lo + per < n ? lo + per : n;
    float s = 0.f;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) s += dz[((i / HW) * C + c) * (long long)HW + i % HW];
    s = block_sum(s, sm);
    if (threadIdx.x == 0) atomicAdd(&db[c], s);
}

// ------------------------------------------------------------------------------------------ conv backward
// dx[b,ci,iy,ix] (+)= sum_{co,kh,kw} dz[b,co,oy,ox] * w[co,ci,kh,kw],  oy*stride - pad + kh == iy
__global__ void __launch_bounds__(256) conv_dgrad_f32_kernel(const float* __restrict__ dz, const float* __restrict__ w,
                                                             float* __restrict__ dx, int cin, int cout, int hin, int win, int hout,
                                                             int wout, int ks, int stride, int accumulate) {
    const int ix = blockIdx.x * 64 + (threadIdx.x & 63);
    const int iy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z / cin, ci = blockIdx.z % cin;
    if (ix >= win || iy >= hin) return;
    const int pad = (ks - 1) / 2;
    float acc = 0.f;
    for (int kh = 0; kh < ks; ++kh) {
        const int ty = iy + pad - kh;
        if (ty < 0 || ty % stride) continue;
        const int oy = ty / stride;
        if (oy >= hout) continue;
        for (int kw = 0; kw < ks; ++kw) {
            const int tx = ix + pad - kw;
            if (tx < 0 || tx % stride) continue;
            const int ox = tx / stride;
            if (ox >= wout) continue;
            const float* dzp = dz + ((size_t)b * cout * hout + oy) * wout + ox;
            const float* wp = w + ((size_t)ci * ks + kh) * ks + kw;
            for (int co = 0; co < cout; ++co) acc = fmaf(dzp[(size_t)co * hout * wout], wp[(size_t)co * cin * ks * ks], acc);
        }
    }
    const size_t o = (((size_t)b * cin + ci) * hin + iy) * win + ix;
    dx[o] = accumulate ? dx[o] + acc : acc;
}

// dw[co,ci,kh,kw] = sum_{b,oy,ox} dz[b,co,oy,ox] * x[b,ci,oy*stride-pad+kh, ox*stride-pad+kw]; one workgroup per (co,ci)
__global__ void __launch_bounds__(256) conv_wgrad_f32_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                             float* __restrict__ dw, int B, int cin, int cout, int hin, int win,
                                                             int hout, int wout, int ks, int stride) {
    __shared__ float sm[4];
    const int ci = blockIdx.x, co = blockIdx.y;
    const int pad = (ks - 1) / 2;
    const long long n = (long long)B * hout * wout;
    float acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = 0.f;
    for (long long i = threadIdx.x; i < n; i += 256) {
        const int ox = (int)(i % wout);
        const int oy = (int)((i / wout) % hout);
        const int b = (int)(i / ((long long)wout * hout));
        const float g = dz[(((size_t)b * cout + co) * hout + oy) * wout + ox];
        const float* xp = x + ((size_t)b * cin + ci) * hin * win;
        for (int kh = 0; kh < ks; ++kh) {
            const int iy = oy * stride - pad + kh;
            if (iy < 0 || iy >= hin) continue;
            for (int kw = 0; kw < ks; ++kw) {
                const int ixx = ox * stride - pad + kw;
                if (ixx < 0 || ixx >= win) continue;
                acc[kh * ks + kw] = fmaf(g, xp[(size_t)iy * win + ixx], acc[kh * ks + kw]);
            }
        }
    }
    for (int t = 0; t < ks * ks; ++t) {
        const float s = block_sum(acc[t], sm);
        if (threadIdx.x == 0) dw[((size_t)co * cin + ci) * ks * ks + t] = s;
    }
}

// ------------------------------------------------------------------------------------------ graph plumbing
__global__ void add_kernel(const float* a, const float* b, float* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = a[i] + b[i];
}
__global__ void accum_kernel(float* dst, const float* src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] += src[i];
}
// out[b, c0 + c, y, x] = src[b, c, y>>up, x>>up]   (route concat / nearest upsample forward)
__global__ void copy_channels_kernel(const float* __restrict__ src, float* __restrict__ out, int B, int csrc, int cout_total, int c0,
                                     int H, int W, int up) {
    const size_t n = (size_t)B * csrc * H * W;
    const int hs = H >> up, ws = W >> up;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        size_t t = i / W;
        const int y = (int)(t % H);
        t /= H;
        const int c = (int)(t % csrc);
        const int b = (int)(t / csrc);
        out[(((size_t)b * cout_total + c0 + c) * H + y) * W + x] = src[(((size_t)b * csrc + c) * hs + (y >> up)) * ws + (x >> up)];
    }
}
// dsrc[b,c,ys,xs] += sum over the (1<<up)^2 children of dout[b, c0+c, y, x]   (route / upsample backward)
__global__ void slice_accum_kernel(const float* __restrict__ dout, float* __restrict__ dsrc, int B, int csrc, int ctotal, int c0,
                                   int H, int W, int up) {
    const int hs = H >> up, ws = W >> up;
    const size_t n = (size_t)B * csrc * hs * ws;
    const int f = 1 << up;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int xs = (int)(i % ws);
        size_t t = i / ws;
        const int ys = (int)(t % hs);
        t /= hs;
        const int c = (int)(t % csrc);
        const int b = (int)(t / csrc);
        float s = 0.f;
        for (int dy = 0; dy < f; ++dy)
            for (int dx = 0; dx < f; ++dx) s += dout[(((size_t)b * ctotal + c0 + c) * H + ys * f + dy) * W + xs * f + dx];
        dsrc[i] += s;
    }
}

// ------------------------------------------------------------------------------------------ YOLO targets + loss
struct YoloGeom {
    int B, A, C, G;
    float aw[8], ah[8];  // anchors / stride (grid units), models.py:123
};

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }
// d BCE(sigmoid(x), t) / dx exactly as autograd composes it: binary_cross_entropy_backward divides by
// max(p(1-p), 1e-12) and sigmoid_backward multiplies by p(1-p) -- equal to (p - t) except where p saturates to 0/1 in
// fp32, where the reference's gradient is 0 (and its loss term is the -100 clamp).
__device__ __forceinline__ float bce_logit_grad(float p, float t) {
    const float pq = (1.0f - p) * p;
    return ((p - t) / fmaxf(pq, 1e-12f)) * pq;
}
__device__ __forceinline__ float wh_iou(float w1, float h1, float w2, float h2) {  // utils/utils.py:193-199
    const float inter = fminf(w1, w2) * fminf(h1, h2);
    return inter / ((w1 * h1 + 1e-16f) + w2 * h2 - inter);
}
__device__ __forceinline__ float iou_cxcywh_p1(float ax, float ay, float aw, float ah, float bx, float by, float bw, float bh) {
    const float ax1 = ax - aw / 2.0f, ax2 = ax + aw / 2.0f, ay1 = ay - ah / 2.0f, ay2 = ay + ah / 2.0f;
    const float bx1 = bx - bw / 2.0f, bx2 = bx + bw / 2.0f, by1 = by - bh / 2.0f, by2 = by + bh / 2.0f;
    const float ix1 = fmaxf(ax1, bx1), iy1 = fmaxf(ay1, by1), ix2 = fminf(ax2, bx2), iy2 = fminf(ay2, by2);
    const float inter = fmaxf(ix2 - ix1 + 1.0f, 0.0f) * fmaxf(iy2 - iy1 + 1.0f, 0.0f);
    const float a1 = (ax2 - ax1 + 1.0f) * (ay2 - ay1 + 1.0f), a2 = (bx2 - bx1 + 1.0f) * (by2 - by1 + 1.0f);
    return inter / (a1 + a2 - inter + 1e-16f);
}

// GIoU box loss (new feature, no reference counterpart -- SURVEY F3): L = 1 - GIoU(pred, target) on corner boxes without the
// +1 rule (same formula as ay_box_iou mode 1), and its gradient with respect to the predicted box (cx, cy, w, h).
__device__ __forceinline__ float giou_loss_grad(float bx, float by, float bw, float bh, float gx, float gy, float gw, float gh,
                                                float* dbox /* [4] or nullptr */) {
    const float x1 = bx - bw / 2.0f, x2 = bx + bw / 2.0f, y1 = by - bh / 2.0f, y2 = by + bh / 2.0f;
    const float X1 = gx - gw / 2.0f, X2 = gx + gw / 2.0f, Y1 = gy - gh / 2.0f, Y2 = gy + gh / 2.0f;
    const float iwr = fminf(x2, X2) - fmaxf(x1, X1), ihr = fminf(y2, Y2) - fmaxf(y1, Y1);
    const float iw = fmaxf(iwr, 0.0f), ih = fmaxf(ihr, 0.0f);
    const float inter = iw * ih;
    const float w1 = x2 - x1, h1 = y2 - y1;
    const float U = w1 * h1 + (X2 - X1) * (Y2 - Y1) - inter + 1e-16f;
    const float cw = fmaxf(x2, X2) - fminf(x1, X1), ch = fmaxf(y2, Y2) - fminf(y1, Y1);
    const float Ch = cw * ch + 1e-16f;
    const float loss = 1.0f - (inter / U - (Ch - U) / Ch);
    if (dbox) {
        // per corner c in (x1, x2, y1, y2): d inter, d area, d hull
        const float di[4] = {(iwr > 0.0f && x1 > X1) ? -ih : 0.0f, (iwr > 0.0f && x2 < X2) ? ih : 0.0f,
                             (ihr > 0.0f && y1 > Y1) ? -iw : 0.0f, (ihr > 0.0f && y2 < Y2) ? iw : 0.0f};
        const float da[4] = {-h1, h1, -w1, w1};
        const float dc[4] = {(x1 < X1) ? -ch : 0.0f, (x2 > X2) ? ch : 0.0f, (y1 < Y1) ? -cw : 0.0f, (y2 > Y2) ? cw : 0.0f};
        float dl[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float dU = da[c] - di[c];
            const float dg = (di[c] * U - inter * dU) / (U * U) + (dU * Ch - U * dc[c]) / (Ch * Ch);
            dl[c] = -dg;
        }
        dbox[0] = dl[0] + dl[1];
        dbox[1] = dl[2] + dl[3];
        dbox[2] = 0.5f * (dl[1] - dl[0]);
        dbox[3] = 0.5f * (dl[3] - dl[2]);
    }
    return loss;
}

// cell state words: bit0 obj, bit1 "noobj cleared" (best anchor or ignore threshold); winner[cell] = last target index
// (the reference's scatter is last-writer-wins in target order: utils/utils.py:310-327).
__global__ void yolo_targets_pass1(const float* __restrict__ tgt, int nT, YoloGeom g, float ignore_thres, int* __restrict__ winner,
                                   unsigned* __restrict__ flags, float* __restrict__ tcls) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nT) return;
    const float* r = tgt + (size_t)t * 6;
    const int b = (int)r[0], label = (int)r[1];
    const float gx = r[2] * g.G, gy = r[3] * g.G, gw = r[4] * g.G, gh = r[5] * g.G;
    int gi = (int)gx, gj = (int)gy;  // trunc like .long(); negative indices wrap like PyTorch indexing
    if (gi < 0) gi += g.G;
    if (gj < 0) gj += g.G;
    if (b < 0 || b >= g.B || gi < 0 || gi >= g.G || gj < 0 || gj >= g.G || label < 0 || label >= g.C) return;
    float best = -1.f;
    int bn = 0;
    for (int a = 0; a < g.A; ++a) {
        const float v = wh_iou(g.aw[a], g.ah[a], gw, gh);
        if (v > best) {
            best = v;
            bn = a;
        }
        if (v > ignore_thres) atomicOr(&flags[((b * g.A + a) * g.G + gj) * g.G + gi], 2u);
    }
    const int cell = ((b * g.A + bn) * g.G + gj) * g.G + gi;
    atomicOr(&flags[cell], 3u);
    atomicMax(&winner[cell], t);
    tcls[(size_t)cell * g.C + label] = 1.0f;
}

// sums: [0] sx [1] sy [2] sw [3] sh [4] conf_obj [5] conf_noobj [6] cls [7] n_obj [8] n_noobj [9] cls_acc_sum
//       [10] conf_obj_sum [11] conf_noobj_sum [12] conf50_sum [13] iou50*det [14] iou75*det
__global__ void yolo_loss_pass(const float* __restrict__ head, const float* __restrict__ tgt, YoloGeom g, const int* __restrict__ winner,
                               const unsigned* __restrict__ flags, const float* __restrict__ tcls, float* __restrict__ sums,
                               float* __restrict__ dhead, int phase, float grad_scale, int box_loss, float* __restrict__ part) {
    // phase 0: accumulate sums (losses un-normalised + counts); phase 1: write dL/dhead using the counts in sums
    const int cells = g.B * g.A * g.G * g.G;
    const int K = 5 + g.C;
    float loc[15];
#pragma unroll
    for (int k = 0; k < 15; ++k) loc[k] = 0.f;
    for (int cell = blockIdx.x * blockDim.x + threadIdx.x; cell < cells; cell += gridDim.x * blockDim.x) {
        const int gi = cell % g.G, gj = (cell / g.G) % g.G, a = (cell / (g.G * g.G)) % g.A, b = cell / (g.G * g.G * g.A);
        const size_t base = (((size_t)b * g.A * K + (size_t)a * K) * g.G + gj) * g.G + gi;  // channel k at + k*G*G
        const size_t cs = (size_t)g.G * g.G;
        const unsigned f = flags[cell];
        const bool obj = f & 1u, noobj = !(f & 2u);
        const float pc = sigm(head[base + 4 * cs]);
        if (phase == 0) {
            loc[12] += pc > 0.5f ? 1.f : 0.f;
            if (noobj) {
                loc[5] += -fmaxf(logf(1.0f - pc), -100.0f);
                loc[8] += 1.f;
                loc[11] += pc;
            }
        } else {
            const float n_noobj = sums[8];
            dhead[base + 4 * cs] = noobj ? grad_scale * 100.0f * bce_logit_grad(pc, 0.0f) / n_noobj : 0.f;
            for (int k = 0; k < K; ++k)
                if (k != 4) dhead[base + k * cs] = 0.f;
        }
        if (!obj) continue;
        const int t = winner[cell];
        const float* r = tgt + (size_t)t * 6;
        const float gx = r[2] * g.G, gy = r[3] * g.G, gw = r[4] * g.G, gh = r[5] * g.G;
        const float tx = gx - floorf(gx), ty = gy - floorf(gy);
        const float tw = logf(gw / g.aw[a] + 1e-16f), th = logf(gh / g.ah[a] + 1e-16f);
        const float px = head[base], py = head[base + cs], pw = head[base + 2 * cs], ph = head[base + 3 * cs];
        const float sx = sigm(px), sy = sigm(py);
        if (phase == 0) {
            if (box_loss == 1) {  // GIoU variant: one term replaces the four squared errors
                loc[0] += giou_loss_grad(sx + gi, sy + gj, expf(pw) * g.aw[a], expf(ph) * g.ah[a], gx, gy, gw, gh, nullptr);
            } else {
                loc[0] += (sx - tx) * (sx - tx);
                loc[1] += (sy - ty) * (sy - ty);
                loc[2] += (pw - tw) * (pw - tw);
                loc[3] += (ph - th) * (ph - th);
            }
            loc[4] += -fmaxf(logf(pc), -100.0f);
            loc[7] += 1.f;
            loc[10] += pc;
            float best = -1.f;
            int arg = 0;
            for (int k = 0; k < g.C; ++k) {
                const float p = sigm(head[base + (5 + k) * cs]);
                const float tc = tcls[(size_t)cell * g.C + k];
                loc[6] += -(tc * fmaxf(logf(p), -100.0f) + (1.0f - tc) * fmaxf(logf(1.0f - p), -100.0f));
                if (p > best) {
                    best = p;
                    arg = k;
                }
            }
            const float cm = (arg == (int)r[1]) ? 1.f : 0.f;
            loc[9] += cm;
            const float iou = iou_cxcywh_p1(sx + gi, sy + gj, expf(pw) * g.aw[a], expf(ph) * g.ah[a], gx, gy, gw, gh);
            const float det = (pc > 0.5f ? 1.f : 0.f) * cm;
            loc[13] += (iou > 0.5f ? 1.f : 0.f) * det;
            loc[14] += (iou > 0.75f ? 1.f : 0.f) * det;
        } else {
            const float n_obj = sums[7];
            const float s = grad_scale / n_obj;
            if (box_loss == 1) {
                const float bw = expf(pw) * g.aw[a], bh = expf(ph) * g.ah[a];
                float db[4];
                giou_loss_grad(sx + gi, sy + gj, bw, bh, gx, gy, gw, gh, db);
                dhead[base] = s * db[0] * sx * (1.0f - sx);   // d(cx)/d(tx) = sigmoid'
                dhead[base + cs] = s * db[1] * sy * (1.0f - sy);
                dhead[base + 2 * cs] = s * db[2] * bw;        // d(w)/d(tw) = w
                dhead[base + 3 * cs] = s * db[3] * bh;
            } else {
                dhead[base] = s * 2.0f * (sx - tx) * sx * (1.0f - sx);
                dhead[base + cs] = s * 2.0f * (sy - ty) * sy * (1.0f - sy);
                dhead[base + 2 * cs] = s * 2.0f * (pw - tw);
                dhead[base + 3 * cs] = s * 2.0f * (ph - th);
            }
            dhead[base + 4 * cs] += s * bce_logit_grad(pc, 1.0f);  // obj cells are never noobj: the += lands on 0
            for (int k = 0; k < g.C; ++k) {
                const float p = sigm(head[base + (5 + k) * cs]);
                dhead[base + (5 + k) * cs] = s * bce_logit_grad(p, tcls[(size_t)cell * g.C + k]) / (float)g.C;
            }
        }
    }
    if (phase == 0) {
        // one partial row per workgroup, added up in a fixed order by yolo_loss_reduce (the per-wave float atomics this replaces --
        // up to 24 576 waves on 4 hot addresses -- were most of the kernel's 250 us, and made the loss scalar depend on the run)
        __shared__ float sm[4][16];
#pragma unroll
        for (int k = 0; k < 15; ++k) {
            const float v = wave_sum(loc[k]);
            if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6][k] = v;
        }
        __syncthreads();
        if (threadIdx.x < 15) part[(size_t)blockIdx.x * 16 + threadIdx.x] = (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
    }
}

// 16 chunks of rows per column, each summed in row order by one thread, then the 16 chunk sums in chunk order: a fixed tree (the
// same bits on every run) of depth n/16 + 16 instead of one chain of n dependent loads (115 us for 1 024 rows)
__global__ void __launch_bounds__(256) yolo_loss_reduce(const float* __restrict__ part, int n_parts, float* __restrict__ sums) {
    __shared__ float sm[16][16];
    const int k = threadIdx.x & 15, c = threadIdx.x >> 4;
    const int per = (n_parts + 15) / 16;
    float s = 0.f;
    if (k < 15)
        for (int w = c * per; w < min((c + 1) * per, n_parts); ++w) s += part[(size_t)w * 16 + k];
    sm[c][k] = s;
    __syncthreads();
    if (threadIdx.x < 15) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += sm[j][threadIdx.x];
        sums[threadIdx.x] = t;
    }
}

// utils/utils.py:276-330 as dense tensors: everything the 10-tuple holds, from the pass-1 cell state (obj / noobj flags,
// last-writer target index, multi-hot classes).  pred_boxes [B,A,G,G,4] (cxcywh, grid units), pred_cls [B,A,G,G,C].
__global__ void build_targets_dense(const float* __restrict__ pred_boxes, const float* __restrict__ pred_cls,
                                    const float* __restrict__ tgt, YoloGeom g, const int* __restrict__ winner,
                                    const unsigned* __restrict__ flags, float* __restrict__ iou_scores,
                                    float* __restrict__ class_mask, uint8_t* __restrict__ obj_mask, uint8_t* __restrict__ noobj_mask,
                                    float* __restrict__ tx, float* __restrict__ ty, float* __restrict__ tw, float* __restrict__ th,
                                    float* __restrict__ tconf) {
    const int cells = g.B * g.A * g.G * g.G;
    for (int cell = blockIdx.x * blockDim.x + threadIdx.x; cell < cells; cell += gridDim.x * blockDim.x) {
        const int a = (cell / (g.G * g.G)) % g.A;
        const unsigned f = flags[cell];
        const bool obj = f & 1u;
        float vx = 0.f, vy = 0.f, vw = 0.f, vh = 0.f, cm = 0.f, iou = 0.f;
        if (obj) {
            const float* r = tgt + (size_t)winner[cell] * 6;
            const float gx = r[2] * g.G, gy = r[3] * g.G, gw = r[4] * g.G, gh = r[5] * g.G;
            vx = gx - floorf(gx);
            vy = gy - floorf(gy);
            vw = logf(gw / g.aw[a] + 1e-16f);
            vh = logf(gh / g.ah[a] + 1e-16f);
            const float* pc = pred_cls + (size_t)cell * g.C;
            float best = pc[0];
            int arg = 0;
            for (int k = 1; k < g.C; ++k)
                if (pc[k] > best) {  // first maximum wins, like argmax
                    best = pc[k];
                    arg = k;
                }
            cm = (arg == (int)r[1]) ? 1.f : 0.f;
            const float* pb = pred_boxes + (size_t)cell * 4;
            iou = iou_cxcywh_p1(pb[0], pb[1], pb[2], pb[3], gx, gy, gw, gh);
        }
        iou_scores[cell] = iou;
        class_mask[cell] = cm;
        obj_mask[cell] = obj ? 1 : 0;
        noobj_mask[cell] = (f & 2u) ? 0 : 1;
        tx[cell] = vx;
        ty[cell] = vy;
        tw[cell] = vw;
        th[cell] = vh;
        tconf[cell] = obj ? 1.f : 0.f;
    }
}

// ------------------------------------------------------------------------------------------ Adam
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ gr, float* __restrict__ m, float* __restrict__ v, size_t n,
                            float lr, float b1, float b2, float eps, float bc1, float bc2, float grad_scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float g = gr[i] * grad_scale;
        const float mi = b1 * m[i] + (1.0f - b1) * g;
        const float vi = b2 * v[i] + (1.0f - b2) * g * g;
        m[i] = mi;
        v[i] = vi;
        // torch.optim.Adam: p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
        p[i] -= (lr / bc1) * mi / (sqrtf(vi) / sqrtf(bc2) + eps);
    }
}

static inline unsigned gridn(size_t n) {
    size_t g = (n + 255) / 256;
    if (g > 32768) g = 32768;
    return (unsigned)(g ? g : 1);
}

}  // namespace ay

using namespace ay;

extern "C" int ay_bn_train_fwd_f32(const float* z, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                   float momentum, float eps, int leaky, float* y, float* save_mean, float* save_invstd, int batch,
                                   int channels, int hw, ay_stream_t stream) {
    AY_CHECK_ARG(z && gamma && beta && running_mean && running_var && y && save_mean && save_invstd && channels > 0, "ay_bn_train_fwd_f32: bad args");
    hipLaunchKernelGGL(bn_train_fwd_kernel, dim3(channels), dim3(256), 0, S(stream), z, gamma, beta, running_mean, running_var, momentum,
                       eps, leaky, y, save_mean, save_invstd, batch, channels, hw);
    AY_CHECK_LAUNCH("bn_train_fwd_kernel");
    return AY_OK;
}

extern "C" int ay_bn_train_bwd_f32(const float* dy, const float* y, const float* z, const float* gamma, const float* save_mean,
                                   const float* save_invstd, int leaky, float* dz, float* dgamma, float* dbeta, int batch,
                                   int channels, int hw, ay_stream_t stream) {
    AY_CHECK_ARG(dy && y && z && gamma && save_mean && save_invstd && dz && dgamma && dbeta, "ay_bn_train_bwd_f32: bad args");
    hipLaunchKernelGGL(bn_train_bwd_kernel, dim3(channels), dim3(256), 0, S(stream), dy, y, z, gamma, save_mean, save_invstd, leaky, dz,
                       dgamma, dbeta, batch, channels, hw);
    AY_CHECK_LAUNCH("bn_train_bwd_kernel");
    return AY_OK;
}

static int bias_grad_launch(const float* dz, float* dbias, int accumulate, bool chunked, int batch, int channels, int hw, ay_stream_t stream) {
    AY_CHECK_ARG(dz && dbias, "ay_bias_grad_f32: null");
    if (!accumulate && hipMemsetAsync(dbias, 0, sizeof(float) * channels, S(stream)) != hipSuccess) {
        set_error("ay_bias_grad_f32: memset failed");
        return AY_ERR_LAUNCH;
    }
    const long long n = (long long)batch * hw;
    int chunks = chunked ? (int)((n + 16383) / 16384) : 1;  // one chunk: one fixed-order sum per channel (reproducible)
    if (chunks > 256) chunks = 256;
    if (chunks < 1) chunks = 1;
    hipLaunchKernelGGL(bias_grad_kernel, dim3(channels, chunks), dim3(256), 0, S(stream), dz, dbias, batch, channels, hw);
    AY_CHECK_LAUNCH("bias_grad_kernel");
    return AY_OK;
}

// the bf16 training step: chunked (up to 256 partial sums per channel, combined by atomics)
extern "C" int ay_bias_grad_f32_acc(const float* dz, float* dbias, int accumulate, int batch, int channels, int hw, ay_stream_t stream) {
    return bias_grad_launch(dz, dbias, accumulate, true, batch, channels, hw, stream);
}

// the fp32 parity path: one workgroup per channel, fixed summation order
extern "C" int ay_bias_grad_f32(const float* dz, float* dbias, int batch, int channels, int hw, ay_stream_t stream) {
    return bias_grad_launch(dz, dbias, 0, false, batch, channels, hw, stream);
}

extern "C" int ay_conv_dgrad_f32(const ay_conv_desc* d, const float* dz, const float* w_oihw, float* dx, int accumulate,
                                 ay_stream_t stream) {
    AY_CHECK_ARG(d && dz && w_oihw && dx, "ay_conv_dgrad_f32: null");
    const long long gz = (long long)d->batch * d->cin;
    AY_CHECK_ARG(gz <= 65535, "ay_conv_dgrad_f32: batch*cin = %lld exceeds grid.z", gz);
    dim3 grid((d->win + 63) / 64, (d->hin + 3) / 4, (unsigned)gz);
    hipLaunchKernelGGL(conv_dgrad_f32_kernel, grid, dim3(256), 0, S(stream), dz, w_oihw, dx, d->cin, d->cout, d->hin, d->win, d->hout,
                       d->wout, d->ksize, d->stride, accumulate);
    AY_CHECK_LAUNCH("conv_dgrad_f32_kernel");
    return AY_OK;
}

extern "C" int ay_conv_wgrad_f32(const ay_conv_desc* d, const float* x, const float* dz, float* dw, ay_stream_t stream) {
    AY_CHECK_ARG(d && x && dz && dw && d->ksize <= 3 && d->cout <= 65535, "ay_conv_wgrad_f32: bad args");
    hipLaunchKernelGGL(conv_wgrad_f32_kernel, dim3(d->cin, d->cout), dim3(256), 0, S(stream), x, dz, dw, d->batch, d->cin, d->cout, d->hin,
                       d->win, d->hout, d->wout, d->ksize, d->stride);
    AY_CHECK_LAUNCH("conv_wgrad_f32_kernel");
    return AY_OK;
}

extern "C" int ay_add_f32(const float* a, const float* b, float* out, size_t n, ay_stream_t stream) {
    AY_CHECK_ARG(a && b && out, "ay_add_f32: null");
    hipLaunchKernelGGL(add_kernel, dim3(gridn(n)), dim3(256), 0, S(stream), a, b, out, n);
    AY_CHECK_LAUNCH("add_kernel");
    return AY_OK;
}

extern "C" int ay_accumulate_f32(float* dst, const float* src, size_t n, ay_stream_t stream) {
    AY_CHECK_ARG(dst && src, "ay_accumulate_f32: null");
    hipLaunchKernelGGL(accum_kernel, dim3(gridn(n)), dim3(256), 0, S(stream), dst, src, n);
    AY_CHECK_LAUNCH("accum_kernel");
    return AY_OK;
}

extern "C" int ay_copy_channels_f32(const float* src, float* out, int batch, int csrc, int ctotal, int c0, int h, int w, int up,
                                    ay_stream_t stream) {
    AY_CHECK_ARG(src && out && c0 >= 0 && c0 + csrc <= ctotal && (up == 0 || up == 1), "ay_copy_channels_f32: bad args");
    hipLaunchKernelGGL(copy_channels_kernel, dim3(gridn((size_t)batch * csrc * h * w)), dim3(256), 0, S(stream), src, out, batch, csrc,
                       ctotal, c0, h, w, up);
    AY_CHECK_LAUNCH("copy_channels_kernel");
    return AY_OK;
}

extern "C" int ay_slice_accumulate_f32(const float* dout, float* dsrc, int batch, int csrc, int ctotal, int c0, int h, int w, int up,
                                       ay_stream_t stream) {
    AY_CHECK_ARG(dout && dsrc && c0 >= 0 && c0 + csrc <= ctotal && (up == 0 || up == 1), "ay_slice_accumulate_f32: bad args");
    hipLaunchKernelGGL(slice_accum_kernel, dim3(gridn((size_t)batch * csrc * (h >> up) * (w >> up))), dim3(256), 0, S(stream), dout, dsrc,
                       batch, csrc, ctotal, c0, h, w, up);
    AY_CHECK_LAUNCH("slice_accum_kernel");
    return AY_OK;
}

constexpr int YOLO_LOSS_WGS = 1024;   // workgroups of the two loss passes (grid-stride over the cells)

extern "C" size_t ay_yolo_loss_workspace_bytes(int batch, int num_anchors, int num_classes, int grid) {
    const size_t cells = (size_t)batch * num_anchors * grid * grid;
    return cells * 4 /*winner*/ + cells * 4 /*flags*/ + cells * num_classes * 4 /*tcls*/ + 64 * 4 /*sums*/ +
           (size_t)YOLO_LOSS_WGS * 16 * 4 /*per-workgroup partial sums*/;
}

static int yolo_loss_impl(const float* head_nchw, const float* targets, int n_targets, int batch, int num_anchors, int num_classes,
                          int grid, int img_dim, const float* anchors_wh, float ignore_thres, float grad_scale, float* dhead,
                          float* sums_out /* device, 16 floats */, void* workspace, size_t workspace_bytes, ay_stream_t stream,
                          int box_loss) {
    AY_CHECK_ARG(head_nchw && anchors_wh && dhead && sums_out && workspace, "ay_yolo_loss_fwd_bwd: null");
    AY_CHECK_ARG(num_anchors > 0 && num_anchors <= 8 && num_classes >= 1 && grid > 0, "ay_yolo_loss_fwd_bwd: bad shape");
    AY_CHECK_ARG(n_targets == 0 || targets, "ay_yolo_loss_fwd_bwd: targets null");
    if (workspace_bytes < ay_yolo_loss_workspace_bytes(batch, num_anchors, num_classes, grid)) {
        set_error("ay_yolo_loss_fwd_bwd: workspace too small");
        return AY_ERR_WORKSPACE;
    }
    hipStream_t st = S(stream);
    YoloGeom g;
    g.B = batch;
    g.A = num_anchors;
    g.C = num_classes;
    g.G = grid;
    const float stride = (float)((double)img_dim / (double)grid);
    for (int a = 0; a < num_anchors; ++a) {
        g.aw[a] = (float)((double)anchors_wh[2 * a] / (double)stride);
        g.ah[a] = (float)((double)anchors_wh[2 * a + 1] / (double)stride);
    }
    const size_t cells = (size_t)batch * num_anchors * grid * grid;
    int* winner = (int*)workspace;
    unsigned* flags = (unsigned*)(winner + cells);
    float* tcls = (float*)(flags + cells);
    float* sums = tcls + cells * num_classes;
    if (hipMemsetAsync(winner, 0xff, cells * 4, st) != hipSuccess || hipMemsetAsync(flags, 0, cells * 4 + cells * num_classes * 4 + 64 * 4, st) != hipSuccess) {
        set_error("ay_yolo_loss_fwd_bwd: memset failed");
        return AY_ERR_LAUNCH;
    }
    if (n_targets > 0) {
        hipLaunchKernelGGL(yolo_targets_pass1, dim3((n_targets + 255) / 256), dim3(256), 0, st, targets, n_targets, g, ignore_thres, winner,
                           flags, tcls);
        AY_CHECK_LAUNCH("yolo_targets_pass1");
    }
    unsigned gr = gridn(cells);
    if (gr > (unsigned)YOLO_LOSS_WGS) gr = YOLO_LOSS_WGS;
    float* part = sums + 64;
    hipLaunchKernelGGL(yolo_loss_pass, dim3(gr), dim3(256), 0, st, head_nchw, targets, g, winner, flags, tcls, sums, dhead, 0, grad_scale, box_loss,
                       part);
    AY_CHECK_LAUNCH("yolo_loss_pass(0)");
    hipLaunchKernelGGL(yolo_loss_reduce, dim3(1), dim3(256), 0, st, part, (int)gr, sums);
    AY_CHECK_LAUNCH("yolo_loss_reduce");
    hipLaunchKernelGGL(yolo_loss_pass, dim3(gr), dim3(256), 0, st, head_nchw, targets, g, winner, flags, tcls, sums, dhead, 1, grad_scale, box_loss,
                       part);
    AY_CHECK_LAUNCH("yolo_loss_pass(1)");
    if (hipMemcpyAsync(sums_out, sums, 16 * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) {
        set_error("ay_yolo_loss_fwd_bwd: copy failed");
        return AY_ERR_LAUNCH;
    }
    return AY_OK;
}

extern "C" int ay_yolo_loss_fwd_bwd(const float* head_nchw, const float* targets, int n_targets, int batch, int num_anchors,
                                    int num_classes, int grid, int img_dim, const float* anchors_wh, float ignore_thres, float grad_scale,
                                    float* dhead, float* sums_out, void* workspace, size_t workspace_bytes, ay_stream_t stream) {
    return yolo_loss_impl(head_nchw, targets, n_targets, batch, num_anchors, num_classes, grid, img_dim, anchors_wh, ignore_thres, grad_scale,
                          dhead, sums_out, workspace, workspace_bytes, stream, 0);
}

extern "C" int ay_yolo_loss_giou_fwd_bwd(const float* head_nchw, const float* targets, int n_targets, int batch, int num_anchors,
                                         int num_classes, int grid, int img_dim, const float* anchors_wh, float ignore_thres,
                                         float grad_scale, float* dhead, float* sums_out, void* workspace, size_t workspace_bytes,
                                         ay_stream_t stream) {
    return yolo_loss_impl(head_nchw, targets, n_targets, batch, num_anchors, num_classes, grid, img_dim, anchors_wh, ignore_thres, grad_scale,
                          dhead, sums_out, workspace, workspace_bytes, stream, 1);
}

extern "C" size_t ay_build_targets_workspace_bytes(int batch, int num_anchors, int grid) {
    return (size_t)batch * num_anchors * grid * grid * 8;  // winner + flags
}

extern "C" int ay_build_targets(const float* pred_boxes, const float* pred_cls, const float* targets, int n_targets, int batch,
                                int num_anchors, int num_classes, int grid, const float* anchors_grid, float ignore_thres,
                                float* iou_scores, float* class_mask, uint8_t* obj_mask, uint8_t* noobj_mask, float* tx, float* ty,
                                float* tw, float* th, float* tcls, float* tconf, void* workspace, size_t workspace_bytes,
                                ay_stream_t stream) {
    AY_CHECK_ARG(pred_boxes && pred_cls && anchors_grid && iou_scores && class_mask && obj_mask && noobj_mask && tx && ty && tw && th &&
                     tcls && tconf && workspace,
                 "ay_build_targets: null");
    AY_CHECK_ARG(num_anchors > 0 && num_anchors <= 8 && num_classes >= 1 && grid > 0 && batch > 0, "ay_build_targets: bad shape");
    AY_CHECK_ARG(n_targets == 0 || targets, "ay_build_targets: targets null");
    if (workspace_bytes < ay_build_targets_workspace_bytes(batch, num_anchors, grid)) {
        set_error("ay_build_targets: workspace too small");
        return AY_ERR_WORKSPACE;
    }
    hipStream_t st = S(stream);
    YoloGeom g;
    g.B = batch;
    g.A = num_anchors;
    g.C = num_classes;
    g.G = grid;
    for (int a = 0; a < num_anchors; ++a) {  // already in grid units (models.py:123), as the reference passes them
        g.aw[a] = anchors_grid[2 * a];
        g.ah[a] = anchors_grid[2 * a + 1];
    }
    const size_t cells = (size_t)batch * num_anchors * grid * grid;
    int* winner = (int*)workspace;
    unsigned* flags = (unsigned*)(winner + cells);
    if (hipMemsetAsync(winner, 0xff, cells * 4, st) != hipSuccess || hipMemsetAsync(flags, 0, cells * 4, st) != hipSuccess ||
        hipMemsetAsync(tcls, 0, cells * num_classes * 4, st) != hipSuccess) {
        set_error("ay_build_targets: memset failed");
        return AY_ERR_LAUNCH;
    }
    if (n_targets > 0) {
        hipLaunchKernelGGL(yolo_targets_pass1, dim3((n_targets + 255) / 256), dim3(256), 0, st, targets, n_targets, g, ignore_thres, winner,
                           flags, tcls);
        AY_CHECK_LAUNCH("yolo_targets_pass1");
    }
    hipLaunchKernelGGL(build_targets_dense, dim3(gridn(cells)), dim3(256), 0, st, pred_boxes, pred_cls, targets, g, winner, flags,
                       iou_scores, class_mask, obj_mask, noobj_mask, tx, ty, tw, th, tconf);
    AY_CHECK_LAUNCH("build_targets_dense");
    return AY_OK;
}

extern "C" int ay_adam_flat(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1,
                            float beta2, float eps, int step, float grad_scale, ay_stream_t stream) {
    AY_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && step >= 1, "ay_adam_flat: bad args");
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adam_kernel, dim3(gridn(n)), dim3(256), 0, S(stream), params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, bc1,
                       bc2, grad_scale);
    AY_CHECK_LAUNCH("adam_kernel");
    return AY_OK;
}
