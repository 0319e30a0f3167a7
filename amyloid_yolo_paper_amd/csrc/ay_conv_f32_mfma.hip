// fp32 parity path on the matrix cores: the convolutional block of the reference (models.py:26-45 executed at :242-248) in the
// reference's own layout -- NCHW fp32 activations, OIHW fp32 filters -- with exact-fp32 MFMA (v_mfma_f32_32x32x2_f32: fp32
// products, fp32 accumulation; 256 FLOP/clk/CU, 157 TFLOP/s on 256 CUs) instead of the VALU loop of ay_conv_f32.hip.
// This is the path whose results meet north_star's parity clause end to end (bit-exact NMS indices, boxes within 1e-4 of the
// reference's CPU run), so it gets a real kernel and a number of its own.
//
// GEMM view per image:  D[co][pixel] = sum_{ci, tap} W[co][ci][tap] * X[ci][pixel + tap];  one MFMA takes K = 2 = the two input
// channels of a pair at one filter tap: lane (c = lane & 31, k = lane >> 5) holds A = W[co0 + c][ci0 + k][tap] and
// B = X[ci0 + k][pixel c + tap].  A workgroup (4 waves) owns 64 output channels x an 8 x 32 pixel tile of one image; a wave
// 64 channels x 2 tile rows (2 x 2 accumulator tiles of 32 x 32).  Per stage of KC input channels the halo tile and the
// filter block are staged global -> registers -> LDS (the loads of stage s+1 are in flight under the MFMAs of stage s; two
// workgroups share a CU and cover each other's barriers):
//   LDS pixels   [kc][IN_H][IN_W] floats, channel stride = 32 mod 64 (stride 2: odd) so the two k-halves of a wave's
//                ds_read_b32 fall on disjoint banks
//   LDS filters  [kc][tap][64 + 1] floats: the on-the-fly transposition OIHW -> [ci][tap][co] writes with a stride of 65
//                floats (conflict-free), the A operand reads 32 consecutive channels
// Route concatenation + nearest x2 upsampling (models.py:86-96, 244-245) are folded into the loader exactly as in the VALU
// kernel: channels [0, cin1) come from src1 (at half resolution when up1), the rest from src2.
// Epilogue in the reference's operation order, unfused (-ffp-contract=off): y = acc * scale + shift; LeakyReLU; + residual.
#include <stdlib.h>

#include "ay_common.h"

namespace ay {

struct F32Args {
    const float* s1;
    const float* s2;
    const float* w;
    const float* scale;
    const float* shift;
    const float* res;
    float* out;
    int cin, cin1, up1, cout, hin, win, hout, wout, leaky;
    int tiles_x, tiles_y, n_cgroups, n_items;
};

constexpr int pad_mod64(int v, int want) {  // smallest value >= v that is `want` modulo 64
    int r = v;
    while (r % 64 != want) ++r;
    return r;
}

template <int KS, int STRIDE, int KC, bool DUAL>
__global__ void __launch_bounds__(256, 2) conv_f32_mfma_kernel(F32Args a) {
    constexpr int BN = 64, TH = 8, TW = 32;
    constexpr int PAD = (KS - 1) / 2, KK2 = KS * KS;
    constexpr int IN_H = (TH - 1) * STRIDE + KS, IN_W = (TW - 1) * STRIDE + KS;
    constexpr int XPL = IN_H * IN_W;
    constexpr int XSTR = STRIDE == 1 ? pad_mod64(XPL, 32) : (XPL | 1);
    constexpr int WT = BN + 1;
    constexpr int WSTR = pad_mod64(KK2 * WT, 32);
    constexpr int NX = (KC * XPL + 255) / 256;       // staged pixels per thread and stage
    constexpr int NW = (BN * KC * KK2 + 255) / 256;  // staged filter taps per thread and stage
    constexpr int LDS_FLOATS = KC * (XSTR + WSTR);
    static_assert(LDS_FLOATS * 4 <= 80 * 1024, "two workgroups per CU");
    static_assert(KC % 2 == 0, "channel pairs");

    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    float* const lx = lds;
    float* const lw = lds + KC * XSTR;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;

    // workgroup -> item: the workgroups of one XCD (id & 7) walk a contiguous range of items, channel groups of a pixel tile
    // adjacent: the input halo tile is fetched from HBM once and then hit in that XCD's L2
    const int per_xcd = (a.n_items + 7) >> 3;
    const int item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || item >= a.n_items) return;
    const int cg = item % a.n_cgroups;
    const int pt = item / a.n_cgroups;
    const int b = pt / (a.tiles_x * a.tiles_y);
    const int y0 = ((pt / a.tiles_x) % a.tiles_y) * TH, x0 = (pt % a.tiles_x) * TW;
    const int co0 = cg * BN;

    // ---- staging maps, fixed for the K loop: element e = i * 256 + tid of the stage's pixel block / filter block ------------
    // Every load is a raw buffer load: a wave-uniform descriptor (rebuilt per stage: base = first channel of the stage, range =
    // what is left of the tensor) + ONE 32-bit per-lane offset that never changes.  Lanes with nothing to fetch (zero padding,
    // channels / filters beyond the tensor) carry an offset beyond every range and read zeros: no 64-bit per-lane addresses, no
    // branches -- with plain pointers the 29 addresses of a stage cost 58 registers and the kernel spilled.
    constexpr unsigned OOB = 0x80000000u;
    const int cin2 = a.cin - a.cin1;
    const int h1 = a.hin >> a.up1, w1 = a.win >> a.up1;
    const unsigned plane1 = (unsigned)(h1 * w1), plane2 = (unsigned)(a.hin * a.win);
    const float* const img1 = a.s1 + (size_t)b * a.cin1 * plane1;
    const float* const img2 = cin2 > 0 ? a.s2 + (size_t)b * cin2 * plane2 : nullptr;
    unsigned xo1[DUAL ? NX : 1], xo2[NX];  // byte offset from the stage's first channel in src1 (DUAL only) / in the plain source
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int e = i * 256 + tid;
        const int kc = e / XPL, P = e % XPL;
        const int iy = y0 * STRIDE - PAD + P / IN_W, ix = x0 * STRIDE - PAD + P % IN_W;
        const bool in = e < KC * XPL && iy >= 0 && iy < a.hin && ix >= 0 && ix < a.win;
        if constexpr (DUAL) {
            xo1[i] = in ? (unsigned)(kc * plane1 + (iy >> a.up1) * w1 + (ix >> a.up1)) * 4u : OOB;
            xo2[i] = in ? (unsigned)(kc * plane2 + iy * a.win + ix) * 4u : OOB;
        } else {  // one source, possibly at half resolution (a lazily upsampled layer)
            xo2[i] = in ? (unsigned)(kc * plane1 + (iy >> a.up1) * w1 + (ix >> a.up1)) * 4u : OOB;
        }
    }
    unsigned wo[NW];  // byte offset of this thread's filter taps from W[0][stage's first channel][0]
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int u = i * 256 + tid;
        const int co = co0 + u / (KC * KK2), j = u % (KC * KK2);
        wo[i] = (u < BN * KC * KK2 && co < a.cout) ? (unsigned)((co * a.cin) * KK2 + j) * 4u : OOB;
    }
    float rx[NX], rw[NW];
    auto issue = [&](int s) __attribute__((always_inline)) {
        const int ci0 = s * KC;
        const bool from1 = ci0 < a.cin1;                      // wave-uniform: a stage never straddles the route boundary (host-checked)
        const float* base = from1 ? img1 + (size_t)ci0 * plane1 : img2 + (size_t)(ci0 - a.cin1) * plane2;
        const int left = from1 ? (a.cin1 - ci0) * (int)plane1 : (a.cin - ci0) * (int)plane2;   // floats up to the end of this source
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, left * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            unsigned off = xo2[i];
            if constexpr (DUAL) off = from1 ? xo1[i] : xo2[i];
            rx[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, off, 0, 0));
        }
        // filters: the range ends with the last filter; a channel tail (cin % KC != 0: the 3-channel stem) is masked by hand, since
        // an offset past the row of one filter lands in the next filter's row, not out of range
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w) + (size_t)ci0 * KK2, 0,
                                                                             (a.cout * a.cin - ci0) * KK2 * 4, 0x00020000);
        const int live = (a.cin - ci0) * KK2;   // taps of a filter row that belong to real channels
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wr, wo[i], 0, 0));
            rw[i] = ((i * 256 + tid) % (KC * KK2) < live) ? v : 0.f;
        }
    };
    auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int e = i * 256 + tid;
            if (e < KC * XPL) lx[(e / XPL) * XSTR + e % XPL] = rx[i];
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int u = i * 256 + tid;
            const int j = u % (KC * KK2);
            if (u < BN * KC * KK2) lw[(j / KK2) * WSTR + (j % KK2) * WT + u / (KC * KK2)] = rw[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    // fragment bases: A = filters of channel c (+32 m), k-half hh; B = pixel (row 2 wave + n, column c), k-half hh
    const float* const fa = lw + hh * WSTR + c;
    const float* const fb = lx + hh * XSTR + (2 * wave * STRIDE) * IN_W + c * STRIDE;

    const int nstages = (a.cin + KC - 1) / KC;
    issue(0);
    commit();
    __syncthreads();
    for (int s = 0; s < nstages; ++s) {
        const bool more = s + 1 < nstages;
        if (more) issue(s + 1);
#pragma unroll
        for (int p = 0; p < KC / 2; ++p) {
#pragma unroll
            for (int tap = 0; tap < KK2; ++tap) {
                const int kh = tap / KS, kw = tap % KS;
                float af[2], bf[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) af[m] = fa[2 * p * WSTR + tap * WT + m * 32];
#pragma unroll
                for (int n = 0; n < 2; ++n) bf[n] = fb[2 * p * XSTR + (n * STRIDE + kh) * IN_W + kw];
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m], bf[n], acc[m][n], 0, 0, 0);
            }
        }
        if (more) {
            __syncthreads();
            commit();
            __syncthreads();
        }
    }

    // ---- epilogue: C/D layout of 32x32: column (pixel) = lane & 31, row (channel) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int ox = x0 + c;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int oy = y0 + 2 * wave + n;
        if (oy >= a.hout || ox >= a.wout) continue;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (co < a.cout) {
                    float y = acc[m][n][r] * a.scale[co] + a.shift[co];
                    if (a.leaky) y = y > 0.f ? y : 0.1f * y;
                    const size_t o = (((size_t)b * a.cout + co) * a.hout + oy) * a.wout + ox;
                    if (a.res) y += a.res[o];
                    a.out[o] = y;
                }
            }
    }
}

template <int KS, int STRIDE, int KC>
static int launch_f32_mfma(const ay_conv_desc* d, const float* src1, int cin1, int up1, const float* src2, const float* w, const float* scale,
                           const float* shift, const float* residual, float* out, hipStream_t st) {
    F32Args a;
    a.s1 = src1, a.s2 = src2, a.w = w, a.scale = scale, a.shift = shift, a.res = residual, a.out = out;
    a.cin = d->cin, a.cin1 = cin1, a.up1 = up1, a.cout = d->cout, a.hin = d->hin, a.win = d->win, a.hout = d->hout, a.wout = d->wout;
    a.leaky = d->leaky;
    a.tiles_x = (d->wout + 31) / 32;
    a.tiles_y = (d->hout + 7) / 8;
    a.n_cgroups = (d->cout + 63) / 64;
    const long long n = (long long)a.tiles_x * a.tiles_y * d->batch * a.n_cgroups;
    if (n <= 0 || n > 0x3fffffffLL) {
        set_error("ay_conv_fwd_f32: grid out of range (%lld)", n);
        return AY_ERR_ARG;
    }
    a.n_items = (int)n;
    const unsigned grid = 8u * (unsigned)((n + 7) / 8);
    if (cin1 < d->cin)
        hipLaunchKernelGGL((conv_f32_mfma_kernel<KS, STRIDE, KC, true>), dim3(grid), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((conv_f32_mfma_kernel<KS, STRIDE, KC, false>), dim3(grid), dim3(256), 0, st, a);
    AY_CHECK_LAUNCH("conv_f32_mfma_kernel");
    return AY_OK;
}

// ay_conv_f32.hip: the shapes of the cfg format (1x1 / 3x3, stride 1 / 2) go to the matrix cores; anything else, and AY_F32_MFMA=0,
// stay on the VALU kernel
int conv_fwd_f32_mfma(const ay_conv_desc* d, const float* src1, int cin1, int up1, const float* src2, const float* w, const float* scale,
                      const float* shift, const float* residual, float* out, hipStream_t st, bool* taken) {
    static const int env = getenv("AY_F32_MFMA") ? atoi(getenv("AY_F32_MFMA")) : 1;
    const int kc = d->ksize == 1 ? 16 : d->stride == 2 ? 4 : 8;
    // buffer descriptors address one image of either source and the filter tensor with 32-bit offsets; a route boundary lies
    // on a stage boundary (the cfgs' routes join 128 / 256-channel tensors)
    const bool fits = (long long)d->cin * d->hin * d->win * 4 < (1ll << 31) && (long long)d->cout * d->cin * d->ksize * d->ksize * 4 < (1ll << 31);
    const bool on = env && fits && (cin1 == d->cin || cin1 % kc == 0);
    *taken = true;
    if (on && d->ksize == 3 && d->stride == 1) return launch_f32_mfma<3, 1, 8>(d, src1, cin1, up1, src2, w, scale, shift, residual, out, st);
    if (on && d->ksize == 3 && d->stride == 2) return launch_f32_mfma<3, 2, 4>(d, src1, cin1, up1, src2, w, scale, shift, residual, out, st);
    if (on && d->ksize == 1 && d->stride == 1) return launch_f32_mfma<1, 1, 16>(d, src1, cin1, up1, src2, w, scale, shift, residual, out, st);
    *taken = false;
    return AY_OK;
}

}  // namespace ay
