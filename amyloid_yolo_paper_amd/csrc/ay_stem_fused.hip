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

__device__ __attribute__((aligned(64))) uint32_t g_zero_page_st[16];
#ifdef AY_PHASE_CLOCK
__device__ unsigned long long g_stem_ticks[8];  // [0] DMA issue, [1] phase B, [2] barrier, [3] phase C, [4] epilogue, [5] tail wait, [6] items
#define STEM_TICK(k)                                   \
    if (wave == 0) {                                   \
        const unsigned long long t_ = wall_clock64();  \
        tk[k] += t_ - tk_last;                         \
        tk_last = t_;                                  \
    }
#else
#define STEM_TICK(k)
#endif

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
    constexpr int IMG_DMAS = (IMG_ELEMS + 63) / 64;       // 60 wave-wide 4-byte DMA instructions per image tile
    constexpr int IMG_BYTES = IMG_DMAS * 256;             // 15360
    constexpr int NDMA = (IMG_DMAS + 15) / 16;            // 4 per wave (waves past the end copy zeros into a scratch line)
    constexpr int NIB = 3;                                // image-tile ring: tiles of items i, i+1, i+2
    constexpr int SLAB = 2 * S_PIXP * 16;                 // one 16-channel chunk of stem output
    constexpr int W1_BYTES = 2 * 9 * 2 * BN * 16;         // 36864
    constexpr int OFF_STEM = NIB * IMG_BYTES + 256;       // + 256-byte scratch line for the padding DMAs
    constexpr int OFF_W1 = OFF_STEM + 2 * SLAB;
    constexpr int OFF_SS0 = OFF_W1 + W1_BYTES;
    constexpr int OFF_SS1 = OFF_SS0 + 256;                // layer-1 [scale 64 | pad to 128][shift] floats (conv_epilogue's LDS form):
    constexpr int LDS_BYTES = OFF_SS1 + 1024;             // no global loads in the loop, whose waits would drain the tile DMAs
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    constexpr int MT = 1, NT = 1;                         // wave tile 32 channels x 32 pixels: waves = 2 (channels) x 8 (rows)

    __shared__ __attribute__((aligned(16))) uint8_t lds[LDS_BYTES];
    __builtin_amdgcn_s_setprio(2);  // above a co-resident merge-NMS wavefront (priority 0)

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
    asm volatile("" ::"v"(wa[0]), "v"(wa[1]));  // complete these loads here: a wait at their first use inside the loop would be
                                                 // executed every item and drain the tile DMAs with it

    // image-tile offset of K index k = 16*ks + 8*hh + e (-1: zero padding of K): a compile-time constant per (ks, e) and lane
    // half, selected by hh at the use (a register table of 16 entries per lane pushed the 128-VGPR budget into scratch, and
    // scratch reloads wait vmcnt(0), which drains the tile DMAs)
    auto koff = [](int k) constexpr { return k < 27 ? ((k / 9) * IH + (k % 9) / 3) * IW + k % 3 : -1; };
    // stem scale/shift: [scale 32][shift 32] floats in LDS (32 registers otherwise; 16 waves leave 128 per lane)
    float* ss0 = reinterpret_cast<float*>(lds + OFF_SS0);
    if (tid < 32) {
        ss0[tid] = s.scale0[tid];
        ss0[32 + tid] = s.shift0[tid];
    }
    if (tid < BN) {
        float* ss1 = reinterpret_cast<float*>(lds + OFF_SS1);
        ss1[tid] = a.scale[tid];
        ss1[128 + tid] = a.shift[tid];
    }

    // image tiles go global -> LDS by 4-byte LDS-DMA (fp32 rows start at arbitrary column offsets), two items ahead, into a ring
    // of three tile buffers: nothing is staged in registers and an HBM round trip has two whole items to complete (with the
    // register prefetch of one item the kernel sat at ~2 us of exposed latency per item).  Pixels outside the image come from a
    // page of zeros; every wave issues NDMA instructions per tile so the waits below are counted.
    const uint8_t* zero_page = reinterpret_cast<const uint8_t*>(g_zero_page_st);
    const unsigned lds_base = lds_addr_of(lds);
    auto issue_image = [&](int it, int slot_i) __attribute__((always_inline)) {
        const int b = it / tiles_per_img;
        const int y0 = ((it / a.tiles_x) % a.tiles_y) * TH, x0 = (it % a.tiles_x) * TW;
        const float* xb = s.x + (size_t)b * 3 * s.H * s.W;
#pragma unroll 1
        for (int i = 0; i < NDMA; ++i) {  // not unrolled: one set of address temporaries
            const int k = i * 16 + wave;  // wave-uniform
            const void* g = zero_page + (lane & 15) * 4;
            int dst = NIB * IMG_BYTES;    // scratch line
            if (k < IMG_DMAS) {
                const int u = k * 64 + lane;
                if (u < IMG_ELEMS) {
                    const int col = u % IW, r = (u / IW) % IH, ci = u / (IW * IH);
                    const int iy = 2 * y0 - 2 + r, ix = 2 * x0 - 2 + col;
                    if (iy >= 0 && iy < s.H && ix >= 0 && ix < s.W) g = xb + ((size_t)ci * s.H + iy) * s.W + ix;
                }
                dst = slot_i * IMG_BYTES + k * 256;
            }
            dma4(g, lds_base + dst);
        }
    };

    // layer-1 fragment addresses (same maps as conv_bf16_ring_kernel with STRIDE 2, NT = 1, WM = 1, WN = 8)
    const int wn = wave & 7, wm = wave >> 3;
    const int pb = (hh * S_PIXP + (wn * 2) * SW + c * 2) * 16;  // pixel (ty = wn, tx = c) -> stem pixel (2ty, 2tx)
    const int wa1 = OFF_W1 + (hh * BN + wm * 32 + c) * 16;

    issue_image(item, 0);
    {
        const int it1 = item + slots;
        if (it1 < last) {
            issue_image(it1, 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();  // tile 0 landed; filters and scale/shift tables written
    int ib = 0;
#ifdef AY_PHASE_CLOCK
    unsigned long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tk_last = wall_clock64();
#endif
    while (true) {
        const int pt = item;
        const int b = pt / tiles_per_img;
        const int y0 = ((pt / a.tiles_x) % a.tiles_y) * TH, x0 = (pt % a.tiles_x) * TW;
        const int next_item = item + slots;
        const bool has_next = next_item < last;
        const bool has_next2 = next_item + slots < last;
        const float* img = reinterpret_cast<const float*>(lds + ib * IMG_BYTES);
        // tile of item i+2 -> the buffer item i-1 read (every wave has passed two barriers since)
        if (has_next2) issue_image(next_item + slots, ib >= 1 ? ib - 1 : NIB - 1);
        STEM_TICK(0)

        // ---- B: stem by MFMA into the slabs ----------------------------------------------------------------
        for (int blk = wave; blk < S_PIXP / 32; blk += 16) {
            const int P = blk * 32 + c;
            const int sy = P / SW, sx = P % SW;  // stem pixel inside the halo tile (P >= S_PIX: padding rows of the slab)
            const bool inside = P < S_PIX;
            const int ibase = inside ? sy * IW + sx : 0;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                unsigned pk[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    // K entries past 27 read the tile buffer's zero tail (filled from the zero page): unconditional loads, the
                    // select is on the index (a load under a lane-dependent condition costs an exec-mask round trip each)
                    auto idx = [&](int k) { return koff(k) >= 0 ? ibase + koff(k) : IMG_ELEMS; };
                    const int i0 = hh ? idx(16 * ks + 8 + 2 * jj) : idx(16 * ks + 2 * jj);
                    const int i1 = hh ? idx(16 * ks + 8 + 2 * jj + 1) : idx(16 * ks + 2 * jj + 1);
                    pk[jj] = pack2bf(img[i0], img[i1]);
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
        STEM_TICK(1)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // raw barriers: __syncthreads would drain the DMAs (vmcnt(0))
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        STEM_TICK(2)

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
        STEM_TICK(3)
        // ---- D: epilogue ----------------------------------------------------------------------------------
        ResRegs<MT, NT> rr;
        conv_epilogue<BN, MT, NT, TW, false, false, false, 2>(a, acc1, rr, b, 0, wm, wn, c, hh, y0, x0,
                                                              reinterpret_cast<const float*>(lds + OFF_SS1));
        STEM_TICK(4)
#ifdef AY_PHASE_CLOCK
        if (wave == 0) ++tk[6];
#endif
        if (!has_next) break;
        // tile i+1 has landed: in issue order this wave's younger operations are the DMAs of tile i+2 (if any) and the two
        // output stores of this item (none if its pixel row lies outside the image); everyone is done reading the slabs
        {
            const bool stored = (y0 + wn) < a.hout;
            if (has_next2) {
                if (stored)
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NDMA + 2) : "memory");
                else
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NDMA) : "memory");
            } else {
                if (stored)
                    asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
                else
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            }
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        STEM_TICK(5)
        item = next_item;
        ib = ib + 1 == NIB ? 0 : ib + 1;
    }
#ifdef AY_PHASE_CLOCK
    if (wave == 0 && lane == 0)
        for (int k = 0; k < 7; ++k) atomicAdd(&g_stem_ticks[k], tk[k]);
#endif
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
    a.deal = nullptr;
    a.canvas_gx = 0;
    a.src1 = nullptr;
    a.c1 = 0;
    const long long n_items = (long long)a.tiles_x * a.tiles_y * batch;
    AY_CHECK_ARG(n_items > 0 && n_items < 0x7fffffffLL, "ay_stem_s2_fused_fwd: grid");
    const int per_xcd = (int)((n_items + 7) / 8);
    const int cu_slots = conv_num_cus() / 8;
    dim3 grid((unsigned)(8 * (per_xcd < cu_slots ? per_xcd : cu_slots)));
    hipLaunchKernelGGL(stem_s2_fused_kernel, grid, dim3(1024), 0, S(stream), s, (int)n_items);
    AY_CHECK_LAUNCH("stem_s2_fused_kernel");
#ifdef AY_PHASE_CLOCK
    if (getenv("AY_DBG") && (atoi(getenv("AY_DBG")) & 8)) {
        unsigned long long t[8] = {0}, z[8] = {0};
        (void)hipStreamSynchronize(S(stream));
        (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_stem_ticks), sizeof(t));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stem_ticks), z, sizeof(z));
        if (t[6])
            fprintf(stderr, "[ay stem] per item (us): dma issue %.2f, stem MFMA %.2f, barrier %.2f, layer-1 MFMA %.2f, epilogue %.2f, tail wait %.2f\n",
                    t[0] * 0.01 / t[6], t[1] * 0.01 / t[6], t[2] * 0.01 / t[6], t[3] * 0.01 / t[6], t[4] * 0.01 / t[6], t[5] * 0.01 / t[6]);
    }
#endif
    return AY_OK;
}
