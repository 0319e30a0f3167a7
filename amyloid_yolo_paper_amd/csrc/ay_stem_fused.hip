// Fused stem: layer 0 (3x3 s1, 3->32, BN, leaky; models.py layer 0) + layer 1 (3x3 s2, 32->64, BN, leaky) in one
// persistent kernel, so the 32-channel full-resolution stem output (4.3 GB at B=64, 1024^2) never goes to HBM.
//
// Per item (8x32 output pixels of layer 1, all 64 channels):
//   A  image halo tile 3 x 19 x 67 fp32 (prefetched one item ahead in registers) -> LDS
//   B  stem on the 17x65 halo tile by MFMA: K = 27 taps*channels padded to 32, image values and stem filters in bf16,
//      fp32 accumulate, fp32 affine + leaky, ONE rounding to bf16, written straight into the [chunk][half][pixel][8]
//      LDS slabs the stride-2 convolution reads (zeros outside the image = layer 1's padding)
//   C  layer 1: 2 chunks x 9 taps of v_mfma_f32_32x32x16_bf16 from those slabs; its filters (36 KiB) stay in LDS for the
//      whole kernel
//   D  conv_epilogue (affine, leaky, bf16, 16-byte stores)
#include "ay_conv_common.h"

namespace ay {

struct StemFusedArgs {
    const float* x;         // [B][3][H][W] f32
    const uint16_t* w0;     // stem filters, bf16 [32 cout][32 k], k = ci*9 + kh*3 + kw (27..31 zero)
    const float* scale0;
    const float* shift0;
    int leaky0;
    int H, W;
    ConvArgs c1;            // layer 1 as a ConvArgs (src unused)
};

// 16 waves: every phase is a chain of dependent LDS round trips per wave, so more (shorter) chains per CU, not wider ones
__global__ void __launch_bounds__(1024) stem_s2_fused_kernel(StemFusedArgs s, int n_items) {
    constexpr int TH = 8, TW = 32, BN = 64;
    constexpr int SH = 2 * TH + 1, SW = 2 * TW + 1;      // 17 x 65 stem pixels feed the tile
    constexpr int S_PIX = SH * SW;                        // 1105
    constexpr int S_PIXP = 1120;                          // padded to 35 blocks of 32
    constexpr int IH = SH + 2, IW = SW + 2;               // 19 x 67 image pixels
    constexpr int IMG_ELEMS = 3 * IH * IW;                // 3819
    constexpr int NT_ = 1024;                                // threads
    constexpr int NIMG = (IMG_ELEMS + NT_ - 1) / NT_;     // 4 prefetch registers per thread
    constexpr int SLAB = 2 * S_PIXP * 16;                 // one 16-channel chunk of stem output
    constexpr int W1_BYTES = 2 * 9 * 2 * BN * 16;         // 36864
    constexpr int OFF_IMG = 0;
    constexpr int OFF_STEM = ((IMG_ELEMS * 4 + 15) / 16) * 16;
    constexpr int OFF_W1 = OFF_STEM + 2 * SLAB;
    constexpr int OFF_SS0 = OFF_W1 + W1_BYTES;
    constexpr int LDS_BYTES = OFF_SS0 + 256;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    constexpr int MT = 1, NT = 1;                         // wave tile 32 channels x 32 pixels: waves = 2 (channels) x 8 (rows)

    __shared__ __attribute__((aligned(16))) uint8_t lds[LDS_BYTES];
    float* img = reinterpret_cast<float*>(lds + OFF_IMG);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, hh = lane >> 5;
    const ConvArgs& a = s.c1;

    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int per_xcd = (n_items + 7) >> 3;
    const int first = xcd * per_xcd;
    const int last = min(first + per_xcd, n_items);
    int item = first + slot;
    if (item >= last) return;
    const int tiles_per_img = a.tiles_x * a.tiles_y;

    // layer-1 filters: resident for the whole kernel
    for (int u = tid; u < W1_BYTES / 16; u += NT_)
        *reinterpret_cast<uint4*>(lds + OFF_W1 + u * 16) = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(a.w) + (size_t)u * 16);
    // stem filters as the MFMA A operand: lane (row co = c, half hh), k-step ks: k = 16*ks + 8*hh + j
    bf16x8 wa[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) wa[ks] = *reinterpret_cast<const bf16x8*>(s.w0 + c * 32 + 16 * ks + 8 * hh);

    // per-lane constants of the stem MFMA: image-tile offset of k = 16*ks + 8*hh + e (-1: zero padding of K), affine
    int koff[2][8];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 16 * ks + 8 * hh + e;
            koff[ks][e] = k < 27 ? ((k / 9) * IH + (k % 9) / 3) * IW + k % 3 : -1;
        }
    // stem scale/shift: [scale 32][shift 32] floats in LDS (32 registers otherwise; 16 waves leave 128 per lane)
    float* ss0 = reinterpret_cast<float*>(lds + OFF_SS0);
    if (tid < 32) {
        ss0[tid] = s.scale0[tid];
        ss0[32 + tid] = s.shift0[tid];
    }

    float rimg[NIMG];
    auto load_image = [&](int it) {
        const int pt = it;  // one channel group: item == pixel tile
        const int b = pt / tiles_per_img;
        const int y0 = ((pt / a.tiles_x) % a.tiles_y) * TH, x0 = (pt % a.tiles_x) * TW;
        const float* xb = s.x + (size_t)b * 3 * s.H * s.W;
#pragma unroll
        for (int i = 0; i < NIMG; ++i) {
            const int u = i * NT_ + tid;
            float v = 0.f;
            if (u < IMG_ELEMS) {
                const int col = u % IW, r = (u / IW) % IH, ci = u / (IW * IH);
                const int iy = 2 * y0 - 2 + r, ix = 2 * x0 - 2 + col;
                if (iy >= 0 && iy < s.H && ix >= 0 && ix < s.W) v = xb[((size_t)ci * s.H + iy) * s.W + ix];
            }
            rimg[i] = v;
        }
    };

    // layer-1 fragment addresses (same maps as conv_bf16_ring_kernel with STRIDE 2, NT = 1, WM = 1, WN = 8)
    const int wn = wave & 7, wm = wave >> 3;
    const int pb = (hh * S_PIXP + (wn * 2) * SW + c * 2) * 16;  // pixel (ty = wn, tx = c) -> stem pixel (2ty, 2tx)
    const int wa1 = OFF_W1 + (hh * BN + wm * 32 + c) * 16;

    load_image(item);
    while (true) {
        const int pt = item;
        const int b = pt / tiles_per_img;
        const int y0 = ((pt / a.tiles_x) % a.tiles_y) * TH, x0 = (pt % a.tiles_x) * TW;
        const int next_item = item + slots;
        const bool has_next = next_item < last;

        // ---- A: image tile -> LDS ----------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < NIMG; ++i) {
            const int u = i * NT_ + tid;
            if (u < IMG_ELEMS) img[u] = rimg[i];
        }
        __syncthreads();
        if (has_next) load_image(next_item);  // in flight during B, C, D

        // ---- B: stem by MFMA into the slabs ----------------------------------------------------------------
        for (int blk = wave; blk < S_PIXP / 32; blk += 16) {
            const int P = blk * 32 + c;
            const int sy = P / SW, sx = P % SW;  // stem pixel inside the halo tile (P >= S_PIX: padding rows of the slab)
            const bool inside = P < S_PIX;
            const float* ip = img + (inside ? sy * IW + sx : 0);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                unsigned pk[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int o0 = koff[ks][2 * jj], o1 = koff[ks][2 * jj + 1];
                    const float v0 = o0 >= 0 ? ip[o0] : 0.f;
                    const float v1 = o1 >= 0 ? ip[o1] : 0.f;
                    pk[jj] = pack2bf(v0, v1);
                }
                const uint4 pv = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[ks], __builtin_bit_cast(bf16x8, pv), acc, 0, 0, 0);
            }
            // rows = stem channels (reg&3)+8*(reg>>2)+4*hh, col = pixel c.  Outside the image the stem output is layer 1's
            // zero padding, not leaky(shift).
            const int gy = 2 * y0 - 1 + sy, gx = 2 * x0 - 1 + sx;
            const bool real = inside && gy >= 0 && gy < s.H && gx >= 0 && gx < s.W;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float o[4];
                const float4 scv = *reinterpret_cast<const float4*>(ss0 + 8 * q + 4 * hh);
                const float4 shv = *reinterpret_cast<const float4*>(ss0 + 32 + 8 * q + 4 * hh);
                const float sc0q[4] = {scv.x, scv.y, scv.z, scv.w}, sh0q[4] = {shv.x, shv.y, shv.z, shv.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float t = acc[4 * q + j] * sc0q[j] + sh0q[j];
                    if (s.leaky0) t = t > 0.f ? t : 0.1f * t;
                    o[j] = real ? t : 0.f;
                }
                // channel 8q+4hh+j -> chunk q>>1, half q&1, element 4hh+j
                uint8_t* dst = lds + OFF_STEM + (q >> 1) * SLAB + ((q & 1) * S_PIXP + P) * 16 + hh * 8;
                *reinterpret_cast<uint2*>(dst) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
            }
        }
        __syncthreads();

        // ---- C: layer 1, 2 chunks x 9 taps ------------------------------------------------------------------
        f32x16 acc1[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[m][0][r] = 0.f;
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kh = tap / 3, kw = tap % 3;
                bf16x8 af[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    af[m] = *reinterpret_cast<const bf16x8*>(lds + wa1 + ((ch * 9 + tap) * 2 * BN + m * 32) * 16);
                const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(lds + OFF_STEM + ch * SLAB + pb + (kh * SW + kw) * 16);
#pragma unroll
                for (int m = 0; m < MT; ++m) acc1[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m], bfr, acc1[m][0], 0, 0, 0);
            }
        }
        // ---- D: epilogue ----------------------------------------------------------------------------------
        ResRegs<MT, NT> rr;
        conv_epilogue<BN, MT, NT, TW, false, false>(a, acc1, rr, b, 0, wm, wn, c, hh, y0, x0);
        if (!has_next) break;
        __syncthreads();  // everyone is done reading the slabs / image before the next item overwrites them
        item = next_item;
    }
}

}  // namespace ay

extern "C" int ay_stem_s2_fused_fwd(const float* x_nchw, const void* stem_w_bf16, const float* scale0, const float* shift0, int leaky0,
                                    const void* w1_packed, const float* scale1, const float* shift1, int leaky1, void* out_blocked,
                                    int batch, int h, int w, ay_stream_t stream) {
    using namespace ay;
    AY_CHECK_ARG(x_nchw && stem_w_bf16 && scale0 && shift0 && w1_packed && scale1 && shift1 && out_blocked, "ay_stem_s2_fused_fwd: null");
    AY_CHECK_ARG(batch > 0 && h > 0 && w > 0 && h % 2 == 0 && w % 2 == 0, "ay_stem_s2_fused_fwd: even image sizes only");
    StemFusedArgs s;
    s.x = x_nchw;
    s.w0 = (const uint16_t*)stem_w_bf16;
    s.scale0 = scale0;
    s.shift0 = shift0;
    s.leaky0 = leaky0;
    s.H = h;
    s.W = w;
    ConvArgs& a = s.c1;
    a.src = nullptr;
    a.w = (const uint8_t*)w1_packed;
    a.scale = scale1;
    a.shift = shift1;
    a.residual = nullptr;
    a.out = (uint8_t*)out_blocked;
    a.batch = batch;
    a.cin = 32;
    a.cout_pad = 64;
    a.hin = h;
    a.win = w;
    a.hout = h / 2;
    a.wout = w / 2;
    a.tiles_x = (a.wout + 31) / 32;
    a.tiles_y = (a.hout + 7) / 8;
    a.n_cgroups = 1;
    a.leaky = leaky1;
    a.dbg = 0;
    a.stagger = 0;
    const long long n_items = (long long)a.tiles_x * a.tiles_y * batch;
    AY_CHECK_ARG(n_items > 0 && n_items < 0x7fffffffLL, "ay_stem_s2_fused_fwd: grid");
    const int per_xcd = (int)((n_items + 7) / 8);
    const int cu_slots = conv_num_cus() / 8;
    dim3 grid((unsigned)(8 * (per_xcd < cu_slots ? per_xcd : cu_slots)));
    hipLaunchKernelGGL(stem_s2_fused_kernel, grid, dim3(1024), 0, S(stream), s, (int)n_items);
    AY_CHECK_LAUNCH("stem_s2_fused_kernel");
    return AY_OK;
}
