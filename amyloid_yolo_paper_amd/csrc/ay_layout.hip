// Weight packing, BN folding, the fp32 stem convolution, route/upsample gather and layout converters.
#include <stdarg.h>

#include "ay_common.h"

namespace ay {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// OIHW f32 -> [cin/16][tap][half][cout_pad][8] bf16
template <typename DT>
__global__ void pack_weights_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, int cout, int cout_pad,
                                    int cin, int ks) {
    DT::enter();
    const int kk2 = ks * ks;
    const size_t total = (size_t)(cin / 16) * kk2 * 2 * cout_pad * 8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % 8);
        size_t t = i / 8;
        const int co = (int)(t % cout_pad);
        t /= cout_pad;
        const int half = (int)(t % 2);
        t /= 2;
        const int tap = (int)(t % kk2);
        const int chunk = (int)(t / kk2);
        const int ci = chunk * 16 + half * 8 + j;
        float v = 0.f;
        if (co < cout) v = w[((size_t)co * cin + ci) * kk2 + tap];
        out[i] = DT::from_f32(v);
    }
}

__global__ void fold_bn_kernel(const float* gamma, const float* beta, const float* mean, const float* var,
                               const float* bias, float eps, float* scale, float* shift, int n, int n_pad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad) return;
    float s = 0.f, t = 0.f;
    if (i < n) {
        if (gamma) {
            s = gamma[i] / sqrtf(var[i] + eps);
            t = beta[i] - mean[i] * s;
        } else {
            s = 1.f;
            t = bias ? bias[i] : 0.f;
        }
    }
    scale[i] = s;
    shift[i] = t;
}

// Stem (models.py layer 0): nchw f32 [B,3,H,W] -> blocked bf16 [B][2][H][W][16], 3x3 s1 pad 1, fp32 math.
// One thread = one output pixel x 32 channels; filters are wave-uniform (scalar loads).
template <typename DT>
__global__ void __launch_bounds__(256) stem_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        uint8_t* __restrict__ out, int H, int W, int leaky) {
    DT::enter();
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
    const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    if (ox >= W || oy >= H) return;
    const size_t plane = (size_t)H * W;
    const float* xb = x + (size_t)b * 3 * plane;
    float in[27];
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iy = oy + kh - 1, ix = ox + kw - 1;
                float v = 0.f;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = xb[ci * plane + (size_t)iy * W + ix];
                in[ci * 9 + kh * 3 + kw] = v;
            }
    const size_t pix = (size_t)oy * W + ox;
#pragma unroll
    for (int half = 0; half < 4; ++half) {  // 8 output channels at a time
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int co = half * 8 + j;
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 27; ++k) s = fmaf(in[k], w[co * 27 + k], s);
            s = s * scale[co] + shift[co];
            if (leaky) s = s > 0.f ? s : 0.1f * s;
            acc[j] = s;
        }
        uint4 o = make_uint4(pack2_scalar<DT>(acc[0], acc[1]), pack2_scalar<DT>(acc[2], acc[3]), pack2_scalar<DT>(acc[4], acc[5]), pack2_scalar<DT>(acc[6], acc[7]));
        const size_t pl = (size_t)b * 2 + (half >> 1);
        *reinterpret_cast<uint4*>(out + (pl * plane + pix) * 32 + (half & 1) * 16) = o;
    }
}

// route + upsample gather: 16-byte units (pixel, half) of the output tensor
__global__ void concat_upsample_kernel(const uint8_t* __restrict__ s1, int p1, int up1, const uint8_t* __restrict__ s2,
                                       int p2, uint8_t* __restrict__ out, int B, int H, int W) {
    const size_t units = (size_t)B * (p1 + p2) * H * W * 2;
    const int H1 = H >> up1, W1 = W >> up1;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < units; i += (size_t)gridDim.x * blockDim.x) {
        const int half = (int)(i & 1);
        size_t t = i >> 1;
        const int x = (int)(t % W);
        t /= W;
        const int y = (int)(t % H);
        t /= H;
        const int pl = (int)(t % (p1 + p2));
        const int b = (int)(t / (p1 + p2));
        uint4 v;
        if (pl < p1) {
            const size_t src = (((size_t)b * p1 + pl) * H1 + (y >> up1)) * W1 + (x >> up1);
            v = *reinterpret_cast<const uint4*>(s1 + src * 32 + half * 16);
        } else {
            const size_t src = (((size_t)b * p2 + (pl - p1)) * H + y) * W + x;
            v = *reinterpret_cast<const uint4*>(s2 + src * 32 + half * 16);
        }
        *reinterpret_cast<uint4*>(out + i * 16) = v;
    }
}

template <typename T, typename DT = Bf16>
__global__ void blocked_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst, int B, int C, int H, int W) {
    DT::enter();
    const size_t total = (size_t)B * C * H * W;
    const int CP = (C + 15) / 16;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        size_t t = i / W;
        const int y = (int)(t % H);
        t /= H;
        const int ch = (int)(t % C);
        const int b = (int)(t / C);
        const size_t s = ((((size_t)b * CP + (ch >> 4)) * H + y) * W + x) * 16 + (ch & 15);
        if constexpr (sizeof(T) == 2)
            dst[i] = DT::to_f32(src[s]);
        else
            dst[i] = src[s];
    }
}

template <typename DT>
__global__ void nchw_to_blocked_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int B, int C, int H,
                                            int W) {
    DT::enter();
    const int CP = (C + 15) / 16;
    const size_t total = (size_t)B * CP * H * W * 16;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 15);
        size_t t = i >> 4;
        const int x = (int)(t % W);
        t /= W;
        const int y = (int)(t % H);
        t /= H;
        const int pl = (int)(t % CP);
        const int b = (int)(t / CP);
        const int ch = pl * 16 + j;
        float v = 0.f;
        if (ch < C) v = src[(((size_t)b * C + ch) * H + y) * W + x];
        dst[i] = DT::from_f32(v);
    }
}

static inline unsigned grid_for(size_t n, int block) {
    size_t g = (n + block - 1) / block;
    if (g > 65536) g = 65536;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace ay

using namespace ay;

extern "C" int ay_version(void) { return AY_ABI_VERSION; }
extern "C" const char* ay_last_error(void) { return ay::g_err; }

extern "C" size_t ay_packed_weight_bytes(int cout_pad, int cin, int ksize) {
    return (size_t)(cin / 16) * ksize * ksize * 2 * cout_pad * 8 * 2;
}

template <typename DT>
static int pack_conv_weights(const float* w_oihw, void* packed, int cout, int cout_pad, int cin, int ksize, ay_stream_t stream) {
    AY_CHECK_ARG(w_oihw && packed && cin % 16 == 0 && cout_pad >= cout && cout_pad % 16 == 0, "ay_pack_conv_weights: bad args");
    const size_t total = ay_packed_weight_bytes(cout_pad, cin, ksize) / 2;
    hipLaunchKernelGGL(pack_weights_kernel<DT>, dim3(grid_for(total, 256)), dim3(256), 0, S(stream), w_oihw, (uint16_t*)packed, cout,
                       cout_pad, cin, ksize);
    AY_CHECK_LAUNCH("pack_weights_kernel");
    return AY_OK;
}
extern "C" int ay_pack_conv_weights_bf16(const float* w_oihw, void* packed, int cout, int cout_pad, int cin, int ksize,
                                         ay_stream_t stream) {
    return pack_conv_weights<Bf16>(w_oihw, packed, cout, cout_pad, cin, ksize, stream);
}
extern "C" int ay_pack_conv_weights_f16(const float* w_oihw, void* packed, int cout, int cout_pad, int cin, int ksize,
                                        ay_stream_t stream) {
    return pack_conv_weights<F16>(w_oihw, packed, cout, cout_pad, cin, ksize, stream);
}

extern "C" int ay_fold_bn(const float* gamma, const float* beta, const float* mean, const float* var, const float* bias,
                          float eps, float* scale, float* shift, int n, int n_pad, ay_stream_t stream) {
    AY_CHECK_ARG(scale && shift && n > 0 && n_pad >= n, "ay_fold_bn: bad args");
    AY_CHECK_ARG(!gamma || (beta && mean && var), "ay_fold_bn: BN needs gamma,beta,mean,var");
    hipLaunchKernelGGL(fold_bn_kernel, dim3((n_pad + 255) / 256), dim3(256), 0, S(stream), gamma, beta, mean, var, bias, eps,
                       scale, shift, n, n_pad);
    AY_CHECK_LAUNCH("fold_bn_kernel");
    return AY_OK;
}

template <typename DT>
static int stem_conv_fwd(const float* x_nchw, const float* w_oihw, const float* scale, const float* shift,
                         void* out_blocked, int batch, int h, int w, int leaky, ay_stream_t stream) {
    AY_CHECK_ARG(x_nchw && w_oihw && scale && shift && out_blocked && batch > 0 && h > 0 && w > 0, "ay_stem_conv_fwd: bad args");
    AY_CHECK_ARG(batch <= 65535, "ay_stem_conv_fwd: batch too large");
    dim3 grid((w + 63) / 64, (h + 3) / 4, batch);
    hipLaunchKernelGGL(stem_conv_kernel<DT>, grid, dim3(256), 0, S(stream), x_nchw, w_oihw, scale, shift, (uint8_t*)out_blocked, h, w,
                       leaky);
    AY_CHECK_LAUNCH("stem_conv_kernel");
    return AY_OK;
}
extern "C" int ay_stem_conv_fwd(const float* x_nchw, const float* w_oihw, const float* scale, const float* shift,
                                void* out_blocked, int batch, int h, int w, int leaky, ay_stream_t stream) {
    return stem_conv_fwd<Bf16>(x_nchw, w_oihw, scale, shift, out_blocked, batch, h, w, leaky, stream);
}
extern "C" int ay_stem_conv_fwd_f16(const float* x_nchw, const float* w_oihw, const float* scale, const float* shift,
                                    void* out_blocked, int batch, int h, int w, int leaky, ay_stream_t stream) {
    return stem_conv_fwd<F16>(x_nchw, w_oihw, scale, shift, out_blocked, batch, h, w, leaky, stream);
}

extern "C" int ay_concat_upsample_bf16(const void* src1, int c1, int up1, const void* src2, int c2, void* out, int batch, int h,
                                       int w, ay_stream_t stream) {
    AY_CHECK_ARG(src1 && out && c1 % 16 == 0 && c2 % 16 == 0 && (c2 == 0 || src2), "ay_concat_upsample_bf16: bad args");
    AY_CHECK_ARG(up1 == 0 || (up1 == 1 && h % 2 == 0 && w % 2 == 0), "ay_concat_upsample_bf16: upsample needs even size");
    const size_t units = (size_t)batch * ((c1 + c2) / 16) * h * w * 2;
    hipLaunchKernelGGL(concat_upsample_kernel, dim3(grid_for(units, 256)), dim3(256), 0, S(stream), (const uint8_t*)src1, c1 / 16,
                       up1, (const uint8_t*)src2, c2 / 16, (uint8_t*)out, batch, h, w);
    AY_CHECK_LAUNCH("concat_upsample_kernel");
    return AY_OK;
}

extern "C" int ay_blocked_bf16_to_nchw_f32(const void* src, float* dst, int batch, int c, int h, int w, ay_stream_t stream) {
    AY_CHECK_ARG(src && dst, "ay_blocked_bf16_to_nchw_f32: null");
    const size_t n = (size_t)batch * c * h * w;
    hipLaunchKernelGGL(blocked_to_nchw_kernel<uint16_t>, dim3(grid_for(n, 256)), dim3(256), 0, S(stream), (const uint16_t*)src, dst,
                       batch, c, h, w);
    AY_CHECK_LAUNCH("blocked_to_nchw_kernel");
    return AY_OK;
}

extern "C" int ay_blocked_f16_to_nchw_f32(const void* src, float* dst, int batch, int c, int h, int w, ay_stream_t stream) {
    AY_CHECK_ARG(src && dst, "ay_blocked_f16_to_nchw_f32: null");
    const size_t n = (size_t)batch * c * h * w;
    hipLaunchKernelGGL((blocked_to_nchw_kernel<uint16_t, F16>), dim3(grid_for(n, 256)), dim3(256), 0, S(stream), (const uint16_t*)src, dst,
                       batch, c, h, w);
    AY_CHECK_LAUNCH("blocked_to_nchw_kernel");
    return AY_OK;
}

extern "C" int ay_blocked_f32_to_nchw_f32(const float* src, float* dst, int batch, int c, int h, int w, ay_stream_t stream) {
    AY_CHECK_ARG(src && dst, "ay_blocked_f32_to_nchw_f32: null");
    const size_t n = (size_t)batch * c * h * w;
    hipLaunchKernelGGL(blocked_to_nchw_kernel<float>, dim3(grid_for(n, 256)), dim3(256), 0, S(stream), src, dst, batch, c, h, w);
    AY_CHECK_LAUNCH("blocked_to_nchw_kernel");
    return AY_OK;
}

extern "C" int ay_nchw_f32_to_blocked_bf16(const float* src, void* dst, int batch, int c, int h, int w, ay_stream_t stream) {
    AY_CHECK_ARG(src && dst, "ay_nchw_f32_to_blocked_bf16: null");
    const size_t n = (size_t)batch * ((c + 15) / 16) * h * w * 16;
    hipLaunchKernelGGL(nchw_to_blocked_bf16_kernel<Bf16>, dim3(grid_for(n, 256)), dim3(256), 0, S(stream), src, (uint16_t*)dst, batch, c, h,
                       w);
    AY_CHECK_LAUNCH("nchw_to_blocked_bf16_kernel");
    return AY_OK;
}

extern "C" int ay_nchw_f32_to_blocked_f16(const float* src, void* dst, int batch, int c, int h, int w, ay_stream_t stream) {
    AY_CHECK_ARG(src && dst, "ay_nchw_f32_to_blocked_f16: null");
    const size_t n = (size_t)batch * ((c + 15) / 16) * h * w * 16;
    hipLaunchKernelGGL(nchw_to_blocked_bf16_kernel<F16>, dim3(grid_for(n, 256)), dim3(256), 0, S(stream), src, (uint16_t*)dst, batch, c, h,
                       w);
    AY_CHECK_LAUNCH("nchw_to_blocked_f16_kernel");
    return AY_OK;
}
