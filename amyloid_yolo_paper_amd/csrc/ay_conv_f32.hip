// fp32 parity path: the same convolutional block in the reference's own layout (nchw f32, OIHW f32
// weights), fp32 accumulation.  Used where results must match the reference's fp32 CPU path
// to 1e-4 (tests, precision="fp32" mode of the Darknet host class).  The cfg format's shapes (1x1 / 3x3, stride 1 / 2) run on
// the exact-fp32 MFMA kernel of ay_conv_f32_mfma.hip; the VALU kernel below serves any other shape and AY_F32_MFMA=0.
// Route concat + nearest x2 upsample (models.py:86-96,244-245) are folded into the loader.
#include "ay_common.h"

namespace ay {

constexpr int CO_T = 4;  // output channels per thread

__global__ void __launch_bounds__(256) conv_f32_kernel(const float* __restrict__ s1, int cin1, int up1,
                                                       const float* __restrict__ s2, const float* __restrict__ w,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const float* __restrict__ res, float* __restrict__ out, int cin,
                                                       int cout, int hin, int win, int hout, int wout, int ks, int stride,
                                                       int leaky) {
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
    const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int cgroups = (cout + CO_T - 1) / CO_T;
    const int b = blockIdx.z / cgroups;
    const int co0 = (blockIdx.z % cgroups) * CO_T;
    if (ox >= wout || oy >= hout) return;
    const int pad = (ks - 1) / 2;
    const int kk2 = ks * ks;
    const int h1 = hin >> up1, w1 = win >> up1;
    const int cin2 = cin - cin1;
    float acc[CO_T];
#pragma unroll
    for (int j = 0; j < CO_T; ++j) acc[j] = 0.f;
    for (int ci = 0; ci < cin; ++ci) {
        const float* plane;
        int sh, pw;
        if (ci < cin1) {
            plane = s1 + ((size_t)b * cin1 + ci) * h1 * w1;
            sh = up1;
            pw = w1;
        } else {
            plane = s2 + ((size_t)b * cin2 + (ci - cin1)) * hin * win;
            sh = 0;
            pw = win;
        }
        for (int kh = 0; kh < ks; ++kh) {
            const int iy = oy * stride - pad + kh;
            if (iy < 0 || iy >= hin) continue;
            for (int kw = 0; kw < ks; ++kw) {
                const int ix = ox * stride - pad + kw;
                if (ix < 0 || ix >= win) continue;
                const float v = plane[(size_t)(iy >> sh) * pw + (ix >> sh)];
#pragma unroll
                for (int j = 0; j < CO_T; ++j) {
                    const int co = co0 + j;
                    if (co < cout) acc[j] = fmaf(v, w[((size_t)co * cin + ci) * kk2 + kh * ks + kw], acc[j]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < CO_T; ++j) {
        const int co = co0 + j;
        if (co >= cout) break;
        float y = acc[j] * scale[co] + shift[co];
        if (leaky) y = y > 0.f ? y : 0.1f * y;
        const size_t o = (((size_t)b * cout + co) * hout + oy) * wout + ox;
        if (res) y += res[o];
        out[o] = y;
    }
}

int conv_fwd_f32_mfma(const ay_conv_desc* d, const float* src1, int cin1, int up1, const float* src2, const float* w, const float* scale,
                      const float* shift, const float* residual, float* out, hipStream_t st, bool* taken);  // ay_conv_f32_mfma.hip

}  // namespace ay

static int conv_fwd_f32(const ay_conv_desc* d, const float* src1, int cin1, int up1, const float* src2,
                        const float* w_oihw, const float* scale, const float* shift, const float* residual, float* out,
                        ay_stream_t stream, bool allow_mfma) {
    using namespace ay;
    AY_CHECK_ARG(d && src1 && w_oihw && scale && shift && out, "ay_conv_fwd_f32: null argument");
    AY_CHECK_ARG(cin1 > 0 && cin1 <= d->cin && (cin1 == d->cin || src2), "ay_conv_fwd_f32: channel split %d/%d", cin1, d->cin);
    AY_CHECK_ARG(up1 == 0 || up1 == 1, "ay_conv_fwd_f32: up1");
    const int pad = (d->ksize - 1) / 2;
    AY_CHECK_ARG(d->hout == (d->hin + 2 * pad - d->ksize) / d->stride + 1 && d->wout == (d->win + 2 * pad - d->ksize) / d->stride + 1,
                 "ay_conv_fwd_f32: output size mismatch");
    if (allow_mfma) {
        bool taken = false;
        const int rc = conv_fwd_f32_mfma(d, src1, cin1, up1, src2, w_oihw, scale, shift, residual, out, S(stream), &taken);
        if (taken) return rc;
    }
    const int cgroups = (d->cout + CO_T - 1) / CO_T;
    const long long gz = (long long)d->batch * cgroups;
    AY_CHECK_ARG(gz <= 65535, "ay_conv_fwd_f32: batch*cout/4 = %lld exceeds grid.z", gz);
    dim3 grid((d->wout + 63) / 64, (d->hout + 3) / 4, (unsigned)gz);
    hipLaunchKernelGGL(conv_f32_kernel, grid, dim3(256), 0, S(stream), src1, cin1, up1, src2, w_oihw, scale, shift, residual, out,
                       d->cin, d->cout, d->hin, d->win, d->hout, d->wout, d->ksize, d->stride, d->leaky);
    AY_CHECK_LAUNCH("conv_f32_kernel");
    return AY_OK;
}

extern "C" int ay_conv_fwd_f32(const ay_conv_desc* d, const float* src1, int cin1, int up1, const float* src2,
                               const float* w_oihw, const float* scale, const float* shift, const float* residual, float* out,
                               ay_stream_t stream) {
    return conv_fwd_f32(d, src1, cin1, up1, src2, w_oihw, scale, shift, residual, out, stream, true);
}

// The same block on the VALU kernel only: one fmaf chain per output in (ci, kh, kw) order.  The fp32 TRAINING engine
// (train_engine.py) keeps this form: its step is pinned element-wise against the reference's own training fixtures
// (tests/golden/train_*.npz), and below a LeakyReLU a gradient comparison is sensitive to the summation order of the forward
// (a pre-activation within 1e-5 of zero takes the other slope; tests/test_gpu_train.py) -- the bars there were set on this order.
extern "C" int ay_conv_fwd_f32_valu(const ay_conv_desc* d, const float* src1, int cin1, int up1, const float* src2,
                                    const float* w_oihw, const float* scale, const float* shift, const float* residual, float* out,
                                    ay_stream_t stream) {
    return conv_fwd_f32(d, src1, cin1, up1, src2, w_oihw, scale, shift, residual, out, stream, false);
}
