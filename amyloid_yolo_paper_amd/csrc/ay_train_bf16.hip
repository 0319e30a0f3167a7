// Training-step kernels of the bf16 MFMA path (activations and activation gradients in blocked bf16, statistics and
// parameter gradients in fp32/fp64): train-mode BatchNorm split into a statistics pass and an apply pass around the MFMA
// convolution (models.py:43-45), its backward, gradient plumbing for shortcut / route / upsample (models.py:86-96,244-248),
// filter re-packing that turns the forward convolution kernel into the data-gradient kernel, and zero insertion for the
// stride-2 data gradient.  All of these are bandwidth-bound elementwise / reduction passes over [B][C/16][H][W][16] bf16.
#include "ay_common.h"

namespace ay {

__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
    const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f[2 * j] = bf2f((uint16_t)(u[j] & 0xffffu));
        f[2 * j + 1] = bf2f((uint16_t)(u[j] >> 16));
    }
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    return make_uint4(pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), pack2bf(f[4], f[5]), pack2bf(f[6], f[7]));
}

// The four BatchNorm passes are HBM streams; each workgroup works inside ONE 16-channel plane of ONE image (grid.y = plane,
// grid.x = image * chunks + chunk), so a thread keeps the parameters of its 8 channels in registers (thread parity = channel
// half: all strides are even), and every thread has UNROLL independent 16-byte loads in flight per operand (with one load per
// loop iteration the passes ran at ~2 TB/s: 41 % of the 1024^2 training step, profiles/r01_train_b16_s1024_kernel_stats.csv).
#ifndef AY_BN_UNROLL
#define AY_BN_UNROLL 4
#endif
#ifndef AY_BN_ROUNDS
#define AY_BN_ROUNDS 8
#endif
constexpr int BN_UNROLL = AY_BN_UNROLL;   // independent 16-byte loads in flight per thread and operand
constexpr int BN_ROUNDS = AY_BN_ROUNDS;   // rounds of BN_UNROLL units per thread and workgroup (see bn_chunks)

// ---- per-channel sums over (B,H,W) of a blocked bf16 tensor: sums[c] += sum z, sums[C + c] += sum z^2 (fp64 atomics, one per
// channel and workgroup after a reduction through LDS); BWD: a = dy, zt = z: sums = (sum dpre, sum dpre * xhat)
template <bool BWD>
__global__ void __launch_bounds__(256) bn_sums_kernel(const uint4* __restrict__ a, const uint4* __restrict__ zt,
                                                      const float* __restrict__ mean, const float* __restrict__ invstd,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta, int leaky,
                                                      double* __restrict__ sums, int C, int HW, int chunks) {
    __shared__ float red[4][2][16];
    const int plane = blockIdx.y;
    const int CP = gridDim.y;
    const int half = threadIdx.x & 1;
    const int c0 = plane * 16 + half * 8;
    float s1[8], s2[8], mu[8], is[8], ga[8], be[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        s1[j] = s2[j] = 0.f;
        if (BWD) {
            const int c = c0 + j < C ? c0 + j : C - 1;
            mu[j] = mean[c];
            is[j] = invstd[c];
            ga[j] = gamma[c];
            be[j] = beta[c];
        }
    }
    const int units = HW * 2;
    const int b = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
    const int stride = chunks * 256;
    const size_t base = ((size_t)b * CP + plane) * units;
    for (int u0 = chunk * 256 + threadIdx.x; u0 < units; u0 += stride * BN_UNROLL) {
        uint4 va[BN_UNROLL], vz[BN_UNROLL];
#pragma unroll
        for (int k = 0; k < BN_UNROLL; ++k) {
            const int u = u0 + k * stride;
            va[k] = make_uint4(0, 0, 0, 0);
            vz[k] = make_uint4(0, 0, 0, 0);
            if (u < units) {
                va[k] = a[base + u];
                if (BWD) vz[k] = zt[base + u];
            }
        }
#pragma unroll
        for (int k = 0; k < BN_UNROLL; ++k) {
            float f[8];
            unpack8(va[k], f);
            if (!BWD) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    s1[j] += f[j];
                    s2[j] += f[j] * f[j];
                }
            } else {  // a zero dy (the fill of an out-of-range unit) adds nothing to either sum
                float z[8];
                unpack8(vz[k], z);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = (z[j] - mu[j]) * is[j];
                    const float pre = xh * ga[j] + be[j];
                    const float d = (leaky && !(pre > 0.f)) ? 0.1f * f[j] : f[j];
                    s1[j] += d;
                    s2[j] += d * xh;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int off = 2; off < 64; off <<= 1) {
            s1[j] += __shfl_xor(s1[j], off);
            s2[j] += __shfl_xor(s2[j], off);
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane < 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            red[wv][0][lane * 8 + j] = s1[j];
            red[wv][1][lane * 8 + j] = s2[j];
        }
    }
    __syncthreads();
    if (threadIdx.x < 32) {  // 16 channels x {sum, sum of squares}
        const int which = threadIdx.x >> 4, ch = threadIdx.x & 15;
        const float v = (red[0][which][ch] + red[1][which][ch]) + (red[2][which][ch] + red[3][which][ch]);
        if (plane * 16 + ch < C) atomicAdd(&sums[which * C + plane * 16 + ch], (double)v);
    }
}

// finalize forward statistics: mean, invstd, running stats (PyTorch semantics)
// y = bf16( leaky(gamma * (z - mean) * invstd + beta) [+ skip] )
// The batch statistics are finalised HERE, from the per-channel sums: every workgroup derives mean / invstd of its 16 channels the
// same way (fp64, then one rounding to fp32), and workgroup 0 of a plane also writes save_mean / save_invstd and updates the running
// statistics (PyTorch momentum semantics, models.py:43) -- the separate one-workgroup finalize launch per layer is gone (72 launches
// of ~5 us per step: small tensors at 416^2 are launch-bound).
__global__ void __launch_bounds__(256) bn_apply_kernel(const uint4* __restrict__ z, const double* __restrict__ sums, double n, float eps,
                                                       float momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
                                                       float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, int leaky,
                                                       const uint4* __restrict__ skip, uint4* __restrict__ y, int C, int CP, int HW, int chunks) {
    const int plane = blockIdx.y;
    const int half = threadIdx.x & 1;
    const int c0 = plane * 16 + half * 8;
    float mu[8], is[8], ga[8], be[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = c0 + j < C ? c0 + j : C - 1;
        const double mean = sums[c] / n;
        double var = sums[C + c] / n - mean * mean;
        if (var < 0.0) var = 0.0;
        mu[j] = (float)mean;
        is[j] = (float)(1.0 / sqrt(var + (double)eps));
        ga[j] = gamma[c];
        be[j] = beta[c];
        if (blockIdx.x == 0 && threadIdx.x < 2 && c0 + j < C) {   // one thread per channel half
            save_mean[c] = mu[j];
            save_invstd[c] = is[j];
            const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
            running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
            running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    }
    const int units = HW * 2;
    const int b = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
    const int stride = chunks * 256;
    const size_t base = ((size_t)b * CP + plane) * units;
    for (int u0 = chunk * 256 + threadIdx.x; u0 < units; u0 += stride * BN_UNROLL) {
        uint4 vz[BN_UNROLL], vs[BN_UNROLL];
#pragma unroll
        for (int k = 0; k < BN_UNROLL; ++k) {
            const int u = u0 + k * stride;
            vz[k] = make_uint4(0, 0, 0, 0);
            vs[k] = make_uint4(0, 0, 0, 0);
            if (u < units) {
                vz[k] = z[base + u];
                if (skip) vs[k] = skip[base + u];
            }
        }
#pragma unroll
        for (int k = 0; k < BN_UNROLL; ++k) {
            const int u = u0 + k * stride;
            float f[8], sk[8];
            unpack8(vz[k], f);
            unpack8(vs[k], sk);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = (f[j] - mu[j]) * is[j] * ga[j] + be[j];
                if (leaky) v = v > 0.f ? v : 0.1f * v;
                if (skip) v += sk[j];
                f[j] = c0 + j < C ? v : 0.f;
            }
            if (u < units) y[base + u] = pack8(f);
        }
    }
}

// dz = gamma*invstd/n * (n*dpre - dbeta - xhat*dgamma), dpre = dy * leaky'(pre)
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const uint4* __restrict__ dy, const uint4* __restrict__ z,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, int leaky,
                                                           const double* __restrict__ sums, float n, uint4* __restrict__ dz, int C, int CP,
                                                           int HW, int chunks, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           int accumulate) {
    const int plane = blockIdx.y;
    const int half = threadIdx.x & 1;
    const int c0 = plane * 16 + half * 8;
    float mu[8], is[8], ga[8], be[8], sb[8], sg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = c0 + j < C ? c0 + j : C - 1;
        mu[j] = mean[c];
        is[j] = invstd[c];
        ga[j] = gamma[c];
        be[j] = beta[c];
        sb[j] = (float)sums[c];
        sg[j] = (float)sums[C + c];
        // the parameter gradients (accumulate != 0: ADDED, gradient accumulation over batches, train.py:116-119): workgroup 0 of the plane
        if (blockIdx.x == 0 && threadIdx.x < 2 && c0 + j < C) {
            dbeta[c] = (accumulate ? dbeta[c] : 0.f) + sb[j];
            dgamma[c] = (accumulate ? dgamma[c] : 0.f) + sg[j];
        }
    }
    const int units = HW * 2;
    const int b = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
    const int stride = chunks * 256;
    const size_t base = ((size_t)b * CP + plane) * units;
    for (int u0 = chunk * 256 + threadIdx.x; u0 < units; u0 += stride * BN_UNROLL) {
        uint4 vd[BN_UNROLL], vz[BN_UNROLL];
#pragma unroll
        for (int k = 0; k < BN_UNROLL; ++k) {
            const int u = u0 + k * stride;
            vd[k] = make_uint4(0, 0, 0, 0);
            vz[k] = make_uint4(0, 0, 0, 0);
            if (u < units) {
                vd[k] = dy[base + u];
                vz[k] = z[base + u];
            }
        }
#pragma unroll
        for (int k = 0; k < BN_UNROLL; ++k) {
            const int u = u0 + k * stride;
            float d[8], zz[8];
            unpack8(vd[k], d);
            unpack8(vz[k], zz);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xh = (zz[j] - mu[j]) * is[j];
                const float pre = xh * ga[j] + be[j];
                const float dp = (leaky && !(pre > 0.f)) ? 0.1f * d[j] : d[j];
                const float k2 = ga[j] * is[j] / n;
                const float v = k2 * (n * dp - sb[j] - xh * sg[j]);
                d[j] = c0 + j < C ? v : 0.f;
            }
            if (u < units) dz[base + u] = pack8(d);
        }
    }
}

// dst += src (bf16, fp32 add, one rounding)
__global__ void accum_bf16_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t units) {
    for (size_t u = (size_t)blockIdx.x * 256 + threadIdx.x; u < units; u += (size_t)gridDim.x * 256) {
        float a[8], b[8];
        unpack8(dst[u], a);
        unpack8(src[u], b);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += b[j];
        dst[u] = pack8(a);
    }
}

// route / upsample backward on blocked tensors: dsrc[b][pl][ys][xs] (+)= sum over the (1<<up)^2 children of dout[b][p0+pl][y][x]
__global__ void slice_accum_bf16_kernel(const uint4* __restrict__ dout, uint4* __restrict__ dsrc, int B, int psrc, int ptot, int p0, int H,
                                        int W, int up, int accumulate) {
    const int hs = H >> up, ws = W >> up, f = 1 << up;
    const size_t units = (size_t)B * psrc * hs * ws * 2;
    for (size_t u = (size_t)blockIdx.x * 256 + threadIdx.x; u < units; u += (size_t)gridDim.x * 256) {
        const int half = (int)(u & 1);
        size_t t = u >> 1;
        const int xs = (int)(t % ws);
        t /= ws;
        const int ys = (int)(t % hs);
        t /= hs;
        const int pl = (int)(t % psrc);
        const int b = (int)(t / psrc);
        float acc[8];
        if (accumulate)
            unpack8(dsrc[u], acc);
        else {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        }
        for (int dy = 0; dy < f; ++dy)
            for (int dx = 0; dx < f; ++dx) {
                float v[8];
                unpack8(dout[((((size_t)b * ptot + p0 + pl) * H + ys * f + dy) * W + xs * f + dx) * 2 + half], v);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += v[j];
            }
        dsrc[u] = pack8(acc);
    }
}

// out[b][pl][2y][2x] = in[b][pl][y][x], zeros elsewhere (stride-2 data gradient as a stride-1 convolution)
__global__ void zero_insert_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, int planes_total, int H, int W, int HO, int WO) {
    const size_t units = (size_t)planes_total * HO * WO * 2;
    for (size_t u = (size_t)blockIdx.x * 256 + threadIdx.x; u < units; u += (size_t)gridDim.x * 256) {
        const int half = (int)(u & 1);
        size_t t = u >> 1;
        const int x = (int)(t % WO);
        t /= WO;
        const int y = (int)(t % HO);
        const size_t pl = t / HO;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (!(x & 1) && !(y & 1) && (y >> 1) < H && (x >> 1) < W) v = in[((pl * H + (y >> 1)) * W + (x >> 1)) * 2 + half];
        out[u] = v;
    }
}

// filters for the data gradient: the forward kernel computes dx = conv(dz, W') with W'[ci][co][kh][kw] = W[co][ci][k-1-kh][k-1-kw];
// packed as [Cout/16 (K chunks)][tap][half][CinPad][8] bf16
__global__ void pack_dgrad_weights_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, int cout, int cout_pad16, int cin,
                                          int cin_pad, int ks) {
    const int kk2 = ks * ks;
    const size_t total = (size_t)(cout_pad16 / 16) * kk2 * 2 * cin_pad * 8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % 8);
        size_t t = i / 8;
        const int ci = (int)(t % cin_pad);
        t /= cin_pad;
        const int half = (int)(t % 2);
        t /= 2;
        const int tap = (int)(t % kk2);
        const int chunk = (int)(t / kk2);
        const int co = chunk * 16 + half * 8 + j;
        const int kh = tap / ks, kw = tap % ks;
        float v = 0.f;
        if (co < cout && ci < cin) v = w[(((size_t)co * cin + ci) * ks + (ks - 1 - kh)) * ks + (ks - 1 - kw)];
        out[i] = f2bf(v);
    }
}

// filter images of the four parity classes of a stride-2 data gradient: [class py*2+px][cout_pad/16][window tap oy*2+ox][half][cin_pad][8];
// window row oy of class py holds filter row kh: py = 0: oy 0 -> kh 1, oy 1 -> none; py = 1: oy 0 -> kh 2, oy 1 -> kh 0 (columns alike)
__global__ void pack_dgrad_s2_weights_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, int cout, int cout_pad, int cin,
                                             int cin_pad) {
    const size_t per_class = (size_t)(cout_pad / 16) * 4 * 2 * cin_pad * 8;
    const size_t total = 4 * per_class;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cls = (int)(i / per_class);
        size_t t = i % per_class;
        const int j = (int)(t % 8);
        t /= 8;
        const int ci = (int)(t % cin_pad);
        t /= cin_pad;
        const int half = (int)(t % 2);
        t /= 2;
        const int tap = (int)(t % 4);
        const int chunk = (int)(t / 4);
        const int co = chunk * 16 + half * 8 + j;
        const int py = cls >> 1, px = cls & 1, oy = tap >> 1, ox = tap & 1;
        const int kh = py ? (oy ? 0 : 2) : (oy ? -1 : 1);
        const int kw = px ? (ox ? 0 : 2) : (ox ? -1 : 1);
        float v = 0.f;
        if (co < cout && ci < cin && kh >= 0 && kw >= 0) v = w[(((size_t)co * cin + ci) * 3 + kh) * 3 + kw];
        out[i] = f2bf(v);
    }
}

// ---- all filter images of a training step in ONE launch.  A step re-packs ~75 forward and ~70 data-gradient images (the weights
// changed); as separate launches that is ~145 kernels of a few microseconds each -- 0.84 ms of a 23-ms step at 416^2 for 0.5 GB of
// traffic.  ay_pack_batch_bf16 walks a job table (device memory, built once per weight layout by the caller): workgroup w takes
// PACK_BLOCK consecutive elements of job work[w].job starting at work[w].first.
constexpr int PACK_BLOCK = 2048;
struct PackJob {             // mirrors ay_pack_job (include/amyloid_yolo.h)
    const float* src;        // OIHW fp32 filters
    uint16_t* dst;           // packed bf16 image
    int32_t kind;            // 0: forward image (ay_pack_conv_weights_bf16), 1: data gradient (ay_pack_dgrad_weights_bf16), 2: stride-2 parity classes
    int32_t cout, cout_pad, cin, cin_pad, ksize;
    uint64_t total;          // elements of dst
};
struct PackWork {
    uint32_t job, first_block;
};

__device__ __forceinline__ uint16_t pack_elem(const PackJob& jb, size_t i) {
    const float* w = jb.src;
    if (jb.kind == 0) {   // [cin/16][tap][half][cout_pad][8]; cin here = channels of the source tensor (multiple of 16)
        const int kk2 = jb.ksize * jb.ksize;
        const int j = (int)(i % 8);
        size_t t = i / 8;
        const int co = (int)(t % jb.cout_pad);
        t /= jb.cout_pad;
        const int half = (int)(t % 2);
        t /= 2;
        const int tap = (int)(t % kk2);
        const int chunk = (int)(t / kk2);
        const int ci = chunk * 16 + half * 8 + j;
        return f2bf(co < jb.cout ? w[((size_t)co * jb.cin + ci) * kk2 + tap] : 0.f);
    }
    if (jb.kind == 1) {   // [cout_pad/16][tap][half][cin_pad][8], flipped taps, transposed channels
        const int ks = jb.ksize, kk2 = ks * ks;
        const int j = (int)(i % 8);
        size_t t = i / 8;
        const int ci = (int)(t % jb.cin_pad);
        t /= jb.cin_pad;
        const int half = (int)(t % 2);
        t /= 2;
        const int tap = (int)(t % kk2);
        const int chunk = (int)(t / kk2);
        const int co = chunk * 16 + half * 8 + j;
        const int kh = tap / ks, kw = tap % ks;
        return f2bf(co < jb.cout && ci < jb.cin ? w[(((size_t)co * jb.cin + ci) * ks + (ks - 1 - kh)) * ks + (ks - 1 - kw)] : 0.f);
    }
    // kind 2: [class][cout_pad/16][window tap][half][cin_pad][8]
    const size_t per_class = (size_t)(jb.cout_pad / 16) * 4 * 2 * jb.cin_pad * 8;
    const int cls = (int)(i / per_class);
    size_t t = i % per_class;
    const int j = (int)(t % 8);
    t /= 8;
    const int ci = (int)(t % jb.cin_pad);
    t /= jb.cin_pad;
    const int half = (int)(t % 2);
    t /= 2;
    const int tap = (int)(t % 4);
    const int chunk = (int)(t / 4);
    const int co = chunk * 16 + half * 8 + j;
    const int py = cls >> 1, px = cls & 1, oy = tap >> 1, ox = tap & 1;
    const int kh = py ? (oy ? 0 : 2) : (oy ? -1 : 1);
    const int kw = px ? (ox ? 0 : 2) : (ox ? -1 : 1);
    return f2bf(co < jb.cout && ci < jb.cin && kh >= 0 && kw >= 0 ? w[(((size_t)co * jb.cin + ci) * 3 + kh) * 3 + kw] : 0.f);
}

__global__ void __launch_bounds__(256) pack_batch_kernel(const PackJob* __restrict__ jobs, const PackWork* __restrict__ work) {
    const PackWork wk = work[blockIdx.x];
    const PackJob jb = jobs[wk.job];
    const size_t base = (size_t)wk.first_block * PACK_BLOCK;
#pragma unroll
    for (int r = 0; r < PACK_BLOCK / 256; ++r) {
        const size_t i = base + (size_t)r * 256 + threadIdx.x;
        if (i < jb.total) jb.dst[i] = pack_elem(jb, i);
    }
}

// chunks of one (image, plane) slice: every thread gets about BN_ROUNDS rounds of BN_UNROLL units where the plane allows, and the
// whole grid stays below ~16k workgroups.  Every workgroup starts by loading its 16 channels' parameters and (the sums kernels)
// ends with a block reduction and one fp64 atomic per channel: with 2 rounds (32 KiB per workgroup) that fixed part was a third of
// a workgroup's life -- 8 rounds: BatchNorm passes 25.5 -> 23.8 ms per step at B=32 / 1024^2 (16 rounds, or 8 loads in flight: the same)
static inline int bn_chunks(int batch, int planes, int HW) {
    const int units = HW * 2;
    int c = (units + 256 * BN_UNROLL * BN_ROUNDS - 1) / (256 * BN_UNROLL * BN_ROUNDS);
    const long long cap = 16384ll / ((long long)batch * planes > 0 ? (long long)batch * planes : 1);
    if (c > cap) c = (int)cap;
    return c < 1 ? 1 : c;
}

static inline unsigned gridu(size_t units) {
    size_t g = (units + 255) / 256;
    if (g > 65535) g = 65535;
    return (unsigned)(g ? g : 1);
}

}  // namespace ay

using namespace ay;

static int bn_train_fwd(const void* z, const float* gamma, const float* beta, float* running_mean, float* running_var,
                        float momentum, float eps, int leaky, const void* skip, void* y, float* save_mean, float* save_invstd,
                        double* sums_ws /* 2*C doubles */, bool ws_is_zero, int batch, int channels, int h, int w, ay_stream_t stream) {
    AY_CHECK_ARG(z && gamma && beta && running_mean && running_var && y && save_mean && save_invstd && sums_ws, "ay_bn_train_fwd_bf16: null");
    hipStream_t st = S(stream);
    const int CP = (channels + 15) / 16, HW = h * w;
    if (!ws_is_zero && hipMemsetAsync(sums_ws, 0, sizeof(double) * 2 * channels, st) != hipSuccess) {
        set_error("memset failed");
        return AY_ERR_LAUNCH;
    }
    const int ch = bn_chunks(batch, CP, HW);
    hipLaunchKernelGGL(bn_sums_kernel<false>, dim3(batch * ch, CP), dim3(256), 0, st, (const uint4*)z, nullptr, nullptr, nullptr, nullptr,
                       nullptr, 0, sums_ws, channels, HW, ch);
    AY_CHECK_LAUNCH("bn_sums_kernel");
    hipLaunchKernelGGL(bn_apply_kernel, dim3(batch * ch, CP), dim3(256), 0, st, (const uint4*)z, sums_ws, (double)batch * HW, eps, momentum,
                       running_mean, running_var, save_mean, save_invstd, gamma, beta, leaky, (const uint4*)skip, (uint4*)y, channels, CP, HW,
                       ch);
    AY_CHECK_LAUNCH("bn_apply_kernel");
    return AY_OK;
}

extern "C" int ay_bn_train_fwd_bf16(const void* z, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                    float momentum, float eps, int leaky, const void* skip, void* y, float* save_mean, float* save_invstd,
                                    double* sums_ws, int batch, int channels, int h, int w, ay_stream_t stream) {
    return bn_train_fwd(z, gamma, beta, running_mean, running_var, momentum, eps, leaky, skip, y, save_mean, save_invstd, sums_ws, false, batch,
                        channels, h, w, stream);
}
// sums_ws holds zeros on entry (the caller clears the workspaces of ALL its layers with one memset per step instead of one per
// layer and pass: 144 small fill launches per training step of Darknet-53)
extern "C" int ay_bn_train_fwd_bf16_zeroed_ws(const void* z, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                              float momentum, float eps, int leaky, const void* skip, void* y, float* save_mean,
                                              float* save_invstd, double* sums_ws_zeroed, int batch, int channels, int h, int w,
                                              ay_stream_t stream) {
    return bn_train_fwd(z, gamma, beta, running_mean, running_var, momentum, eps, leaky, skip, y, save_mean, save_invstd, sums_ws_zeroed, true,
                        batch, channels, h, w, stream);
}

// the same layer when the producer of z already left (sum z, sum z^2) in sums_ws (ay_stem_train_fwd_stats_bf16): the apply pass alone
extern "C" int ay_bn_train_apply_bf16(const void* z, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                      float momentum, float eps, int leaky, const void* skip, void* y, float* save_mean, float* save_invstd,
                                      const double* sums_ws /* 2*C doubles, filled */, int batch, int channels, int h, int w, ay_stream_t stream) {
    AY_CHECK_ARG(z && gamma && beta && running_mean && running_var && y && save_mean && save_invstd && sums_ws, "ay_bn_train_apply_bf16: null");
    const int CP = (channels + 15) / 16, HW = h * w;
    const int ch = bn_chunks(batch, CP, HW);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(batch * ch, CP), dim3(256), 0, S(stream), (const uint4*)z, sums_ws, (double)batch * HW, eps, momentum,
                       running_mean, running_var, save_mean, save_invstd, gamma, beta, leaky, (const uint4*)skip, (uint4*)y, channels, CP, HW,
                       ch);
    AY_CHECK_LAUNCH("bn_apply_kernel");
    return AY_OK;
}

static int bn_train_bwd(const void* dy, const void* z, const float* gamma, const float* beta, const float* save_mean,
                        const float* save_invstd, int leaky, void* dz, float* dgamma, float* dbeta, double* sums_ws, bool ws_is_zero,
                        int accumulate, int batch, int channels, int h, int w, ay_stream_t stream) {
    AY_CHECK_ARG(dy && z && gamma && beta && save_mean && save_invstd && dz && dgamma && dbeta && sums_ws, "ay_bn_train_bwd_bf16: null");
    hipStream_t st = S(stream);
    const int CP = (channels + 15) / 16, HW = h * w;
    if (!ws_is_zero && hipMemsetAsync(sums_ws, 0, sizeof(double) * 2 * channels, st) != hipSuccess) {
        set_error("memset failed");
        return AY_ERR_LAUNCH;
    }
    const int ch = bn_chunks(batch, CP, HW);
    hipLaunchKernelGGL(bn_sums_kernel<true>, dim3(batch * ch, CP), dim3(256), 0, st, (const uint4*)dy, (const uint4*)z, save_mean, save_invstd,
                       gamma, beta, leaky, sums_ws, channels, HW, ch);
    AY_CHECK_LAUNCH("bn_sums_kernel<bwd>");
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(batch * ch, CP), dim3(256), 0, st, (const uint4*)dy, (const uint4*)z, save_mean,
                       save_invstd, gamma, beta, leaky, sums_ws, (float)((double)batch * HW), (uint4*)dz, channels, CP, HW, ch, dgamma, dbeta,
                       accumulate);
    AY_CHECK_LAUNCH("bn_bwd_apply_kernel");
    return AY_OK;
}

extern "C" int ay_bn_train_bwd_bf16_acc(const void* dy, const void* z, const float* gamma, const float* beta, const float* save_mean,
                                        const float* save_invstd, int leaky, void* dz, float* dgamma, float* dbeta, double* sums_ws,
                                        int accumulate, int batch, int channels, int h, int w, ay_stream_t stream) {
    return bn_train_bwd(dy, z, gamma, beta, save_mean, save_invstd, leaky, dz, dgamma, dbeta, sums_ws, false, accumulate, batch, channels, h, w, stream);
}
extern "C" int ay_bn_train_bwd_bf16_acc_zeroed_ws(const void* dy, const void* z, const float* gamma, const float* beta, const float* save_mean,
                                                  const float* save_invstd, int leaky, void* dz, float* dgamma, float* dbeta,
                                                  double* sums_ws_zeroed, int accumulate, int batch, int channels, int h, int w,
                                                  ay_stream_t stream) {
    return bn_train_bwd(dy, z, gamma, beta, save_mean, save_invstd, leaky, dz, dgamma, dbeta, sums_ws_zeroed, true, accumulate, batch, channels, h, w, stream);
}

extern "C" int ay_bn_train_bwd_bf16(const void* dy, const void* z, const float* gamma, const float* beta, const float* save_mean,
                                    const float* save_invstd, int leaky, void* dz, float* dgamma, float* dbeta, double* sums_ws, int batch,
                                    int channels, int h, int w, ay_stream_t stream) {
    return ay_bn_train_bwd_bf16_acc(dy, z, gamma, beta, save_mean, save_invstd, leaky, dz, dgamma, dbeta, sums_ws, 0, batch, channels, h, w,
                                    stream);
}

extern "C" int ay_accumulate_bf16(void* dst, const void* src, size_t n_elems, ay_stream_t stream) {
    AY_CHECK_ARG(dst && src && n_elems % 8 == 0, "ay_accumulate_bf16: bad args");
    hipLaunchKernelGGL(accum_bf16_kernel, dim3(gridu(n_elems / 8)), dim3(256), 0, S(stream), (uint4*)dst, (const uint4*)src, n_elems / 8);
    AY_CHECK_LAUNCH("accum_bf16_kernel");
    return AY_OK;
}

extern "C" int ay_slice_accumulate_bf16(const void* dout, void* dsrc, int batch, int csrc, int ctotal, int c0, int h, int w, int up,
                                        int accumulate, ay_stream_t stream) {
    AY_CHECK_ARG(dout && dsrc && csrc % 16 == 0 && ctotal % 16 == 0 && c0 % 16 == 0 && (up == 0 || up == 1), "ay_slice_accumulate_bf16: bad args");
    const size_t units = (size_t)batch * (csrc / 16) * (h >> up) * (w >> up) * 2;
    hipLaunchKernelGGL(slice_accum_bf16_kernel, dim3(gridu(units)), dim3(256), 0, S(stream), (const uint4*)dout, (uint4*)dsrc, batch, csrc / 16,
                       ctotal / 16, c0 / 16, h, w, up, accumulate);
    AY_CHECK_LAUNCH("slice_accum_bf16_kernel");
    return AY_OK;
}

extern "C" int ay_zero_insert_bf16(const void* in, void* out, int batch, int channels, int h, int w, int ho, int wo, ay_stream_t stream) {
    AY_CHECK_ARG(in && out && channels % 16 == 0 && ho >= 2 * h - 1 && wo >= 2 * w - 1, "ay_zero_insert_bf16: bad args");
    const int planes = batch * (channels / 16);
    const size_t units = (size_t)planes * ho * wo * 2;
    hipLaunchKernelGGL(zero_insert_kernel, dim3(gridu(units)), dim3(256), 0, S(stream), (const uint4*)in, (uint4*)out, planes, h, w, ho, wo);
    AY_CHECK_LAUNCH("zero_insert_kernel");
    return AY_OK;
}

extern "C" int ay_pack_batch_block(void) { return ay::PACK_BLOCK; }

extern "C" int ay_pack_batch_bf16(const void* jobs_device, const void* work_device, int n_work, ay_stream_t stream) {
    static_assert(sizeof(PackJob) == 48 && sizeof(PackWork) == 8, "ay_pack_job / ay_pack_work layout");
    AY_CHECK_ARG(jobs_device && work_device && n_work > 0, "ay_pack_batch_bf16: bad args");
    hipLaunchKernelGGL(pack_batch_kernel, dim3((unsigned)n_work), dim3(256), 0, S(stream), (const PackJob*)jobs_device, (const PackWork*)work_device);
    AY_CHECK_LAUNCH("pack_batch_kernel");
    return AY_OK;
}

extern "C" size_t ay_packed_dgrad_s2_weight_bytes(int cout_pad, int cin_pad) {
    return (size_t)4 * (cout_pad / 16) * 4 * 2 * cin_pad * 8 * 2;
}

extern "C" int ay_pack_dgrad_s2_weights_bf16(const float* w_oihw, void* packed, int cout, int cout_pad, int cin, int cin_pad, ay_stream_t stream) {
    AY_CHECK_ARG(w_oihw && packed && cin_pad >= cin && cin_pad % 32 == 0 && cout_pad >= cout && cout_pad % 16 == 0, "ay_pack_dgrad_s2_weights_bf16: bad args");
    const size_t total = (size_t)4 * (cout_pad / 16) * 4 * 2 * cin_pad * 8;
    hipLaunchKernelGGL(pack_dgrad_s2_weights_kernel, dim3(gridu(total)), dim3(256), 0, S(stream), w_oihw, (uint16_t*)packed, cout, cout_pad, cin,
                       cin_pad);
    AY_CHECK_LAUNCH("pack_dgrad_s2_weights_kernel");
    return AY_OK;
}

extern "C" int ay_pack_dgrad_weights_bf16(const float* w_oihw, void* packed, int cout, int cin, int cin_pad, int ksize, ay_stream_t stream) {
    AY_CHECK_ARG(w_oihw && packed && cin_pad >= cin && cin_pad % 32 == 0, "ay_pack_dgrad_weights_bf16: bad args");
    const int cout_pad16 = (cout + 15) / 16 * 16;
    const size_t total = (size_t)(cout_pad16 / 16) * ksize * ksize * 2 * cin_pad * 8;
    hipLaunchKernelGGL(pack_dgrad_weights_kernel, dim3(gridu(total)), dim3(256), 0, S(stream), w_oihw, (uint16_t*)packed, cout, cout_pad16, cin,
                       cin_pad, ksize);
    AY_CHECK_LAUNCH("pack_dgrad_weights_kernel");
    return AY_OK;
}
