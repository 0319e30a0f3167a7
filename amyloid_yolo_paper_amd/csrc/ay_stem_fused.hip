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
    unsigned m_tx, m_tpi;   // v2: floor(2^32 / tiles_x), floor(2^32 / tiles per image) (0xffffffff for a divisor of 1)
    ConvArgs c1;            // layer 1 as a ConvArgs (src unused)
};

// 16 waves: every phase is a chain of dependent LDS round trips per wave, so more (shorter) chains per CU, not wider ones
template <typename DT>
__global__ void __launch_bounds__(1024) stem_s2_fused_kernel(StemFusedArgs s, int n_items) {
    DT::enter();
    typedef typename DT::vec8 vec8;
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
    vec8 wa[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) wa[ks] = *reinterpret_cast<const vec8*>(s.w0 + c * 32 + 16 * ks + 8 * hh);
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
                    pk[jj] = pack2_scalar<DT>(img[i0], img[i1]);
                }
                const uint4 pv = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                acc = DT::mfma32(wa[ks], __builtin_bit_cast(vec8, pv), acc);
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
                *reinterpret_cast<uint2*>(dst) = make_uint2(pack2_scalar<DT>(o[0], o[1]), pack2_scalar<DT>(o[2], o[3]));
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
                vec8 af[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    af[m] = *reinterpret_cast<const vec8*>(lds + wa1 + ((ch * 9 + tap) * 2 * BN + m * 32) * 16);
                const vec8 bfr = *reinterpret_cast<const vec8*>(lds + OFF_STEM + ch * SLAB + pb + (kh * SW + kw) * 16);
#pragma unroll
                for (int m = 0; m < MT; ++m) acc1[m][0] = DT::mfma32(af[m], bfr, acc1[m][0]);
            }
        }
        STEM_TICK(3)
        // ---- D: epilogue ----------------------------------------------------------------------------------
        ResRegs<MT, NT> rr;
        conv_epilogue<BN, MT, NT, TW, false, false, false, 2, false, false, DT>(a, acc1, rr, b, 0, wm, wn, c, hh, y0, x0,
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

// ---------------------------------------------------------------------------------------------------------------------
// v2: the same two layers, phases PIPELINED across two wave groups, with the LDS traffic and the instruction count cut.
// v1's clock per item (8x32 outputs): tile DMA issue 0.9 us, stem phase 2.5 + 1.9 us at its barrier, layer-1 MFMA 1.2, epilogue
// 0.5, tail 0.7 -- 7.8 us against 1.2 us of MFMA work.  What binds it is instruction issue (16 waves x ~1200 instructions per
// item: per-lane 64-bit DMA addresses, item coordinates by integer division in every phase, single-value converts,
// compare+select activations) and the LDS pipe (every layer-1 MFMA reads 2 KB of fragments: 32x32 register tiles have no
// reuse; stride-2 pixel reads of 16-byte cells are 2-way bank conflicts).  Here:
//   * an item is 4x32 outputs; the stem output has TWO slab sets in LDS; waves 0-7 (producers) compute the stem of item k into one
//     set while waves 8-15 (consumers) run layer 1 + epilogue of item k-1 from the other; one barrier per item.  The two roles are
//     separate loops, so each keeps its own constants in registers:
//   * a consumer wave holds its 18 filter fragments of layer 1 (72 VGPRs) for the whole kernel: only pixel fragments come from LDS
//   * stem slabs keep even and odd pixel columns apart, so the stride-2 reads of layer 1 are contiguous 16-byte cells
//   * the image tile arrives by 16-byte buffer DMA (11 wave instructions per tile instead of 35 of 4 bytes; lanes outside the
//     image read out of the descriptor's range = zeros): the window starts 4 columns left of the tile so that every piece is
//     16-byte aligned in HBM (needs W % 4 == 0; other widths take v1).  Item coordinates come from one multiply-high division per
//     item, computed when its tile is issued and kept in a register queue for the two later iterations that use them
//   * the tile is stored [row][channel][column]: the 27 taps of a stem pixel are a 9 x 3 grid (rho = 3 kh + ci, kw) in it, and
//     the K = 32 slots are ORDERED so that the 8 taps of lane half 1 are those of half 0 moved down 3 (K-step 0) or 2 (K-step
//     1) grid rows: every LDS read of the stem is one per-lane base + an immediate; the 5 spare slots carry zero filters and
//     re-read taps of the same or the next pixel row (finite whenever the image is).  Producers hold scale / shift in
//     registers, LeakyReLU is max(t, slope t), two values per convert
template <typename DT>
__global__ void __launch_bounds__(1024) stem_s2_fused_v2_kernel(StemFusedArgs s, int n_items) {
    DT::enter();
    typedef typename DT::vec8 vec8;
    constexpr int TH = 4, TW = 32, BN = 64;
    constexpr int SH = 2 * TH + 1, SW = 2 * TW + 1;      // 9 x 65 stem pixels feed the tile
    constexpr int S_PIX = SH * SW;                        // 585
    constexpr int NBLK = (S_PIX + 31) / 32;               // 19 blocks of 32 stem pixels
    constexpr int HALFW = TW + 1;                         // 33 even (and up to 33 odd) columns per slab row
    constexpr int ROWP = 2 * HALFW;                       // slab row: [even columns 33][odd columns 33] cells of 16 B
    constexpr int S_POS = SH * ROWP;                      // 594 cells; the lanes of the last block beyond pixel 584 write behind them
    constexpr int S_POSP = S_POS + NBLK * 32 - S_PIX;     // 617
    constexpr int IH = SH + 3;                            // 12 image rows 2*y0-2 .. (11 used + 1 that only zero filters see)
    constexpr int IW = 72;                                // image columns 2*x0-4 .. 2*x0+67 (used: 2*x0-2 .. 2*x0+64)
    constexpr int XOFF = 2;                               // window column of image column 2*x0-2
    constexpr int ROW_PIECES = IW / 4;                    // 16-byte pieces per row
    constexpr int IMG_PIECES = IH * 3 * ROW_PIECES;       // 648: [row][channel][column piece]
    constexpr int IMG_DMAS = (IMG_PIECES + 63) / 64;      // 11 wave-wide DMA instructions per tile: one per wave (waves 11-15: none)
    constexpr int IMG_BYTES = IMG_DMAS * 1024;
    constexpr int NIB = 3;
    constexpr int SLAB = 2 * S_POSP * 16;                 // one 16-channel chunk of stem output: [half][cell][8 ch]
    constexpr int SET = 2 * SLAB;                         // both chunks
    constexpr int OFF_SCRATCH = NIB * IMG_BYTES;          // where the DMA of a wave without a piece lands
    constexpr int OFF_STEM = OFF_SCRATCH + 1024;
    constexpr int OFF_SS1 = OFF_STEM + 2 * SET;
    constexpr int LDS_BYTES = OFF_SS1 + 1024;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static_assert(IMG_DMAS <= 16, "one tile DMA per wave");
    constexpr int MT = 1, NT = 1;
    constexpr unsigned OOB = 0x80000000u;

    __shared__ __attribute__((aligned(16))) uint8_t lds[LDS_BYTES];
    __builtin_amdgcn_s_setprio(2);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, hh = lane >> 5;
    const ConvArgs& a = s.c1;

    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int per_xcd = (n_items + 7) >> 3;
    const int first = xcd * per_xcd;
    const int last = min(first + per_xcd, n_items);
    if (first + slot >= last) return;
    const int nk = (last - (first + slot) + slots - 1) / slots;   // items of this workgroup: first + slot + k * slots, k < nk
    const unsigned tiles_per_img = (unsigned)(a.tiles_x * a.tiles_y);
    struct Coord { int b, y0, x0; };
    auto coord_of = [&](int k) __attribute__((always_inline)) {   // item k of this workgroup -> image, tile origin
        const unsigned it = (unsigned)(first + slot + k * slots);
        unsigned b = __umulhi(it, s.m_tpi), r = it - b * tiles_per_img;
        if (r >= tiles_per_img) ++b, r -= tiles_per_img;
        unsigned ty = __umulhi(r, s.m_tx), tx = r - ty * (unsigned)a.tiles_x;
        if (tx >= (unsigned)a.tiles_x) ++ty, tx -= (unsigned)a.tiles_x;
        return Coord{(int)b, (int)ty * TH, (int)tx * TW};
    };

    if (tid < BN) {
        float* ss1 = reinterpret_cast<float*>(lds + OFF_SS1);
        ss1[tid] = a.scale[tid];
        ss1[128 + tid] = a.shift[tid];
    }
    const unsigned lds_base = lds_addr_of(lds);
    // this lane's piece of a tile: piece u = wave*64 + lane -> (row, channel, 4 columns)
    const int pu = wave * 64 + lane;
    const bool pu_ok = wave < IMG_DMAS && pu < IMG_PIECES;
    const int pu_col = (pu % ROW_PIECES) * 4, pu_ci = (pu / ROW_PIECES) % 3, pu_row = pu / (ROW_PIECES * 3);
    const int plane_elems = s.H * s.W;
    const int pu_gofs = pu_ci * plane_elems + pu_row * s.W + pu_col;   // relative to image pixel (2*y0-2, 2*x0-4) of channel 0
    auto issue_image = [&](const Coord& t, int slot_i) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(s.x) + (size_t)t.b * 3 * plane_elems, 0, 3 * plane_elems * 4, 0x00020000);
        const int ybase = 2 * t.y0 - 2, xbase = 2 * t.x0 - 4;
        const bool ok = pu_ok && (unsigned)(ybase + pu_row) < (unsigned)s.H && (unsigned)(xbase + pu_col) < (unsigned)s.W;
        const unsigned vo = ok ? (unsigned)((pu_gofs + ybase * s.W + xbase) * 4) : OOB;
        dma16_buf(rs, vo, 0u, lds_base + (wave < IMG_DMAS ? slot_i * IMG_BYTES + wave * 1024 : OFF_SCRATCH));
    };

    // ---- prologue: tiles of items 0 and 1 in flight, tile 0 landed.  Coordinate queue: cA = item k, cB = item k+1 at loop entry
    Coord cA = coord_of(0), cB = cA;
    issue_image(cA, 0);
    if (nk > 1) {
        cB = coord_of(1);
        issue_image(cB, 1);
        asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
#ifdef AY_PHASE_CLOCK
    unsigned long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tk_last = wall_clock64();
#define STEM2_TICK(w0, kk)                                   \
    if (wave == w0) {                                        \
        const unsigned long long t_ = wall_clock64();        \
        tk[kk] += t_ - tk_last;                              \
        tk_last = t_;                                        \
    }
#else
#define STEM2_TICK(w0, kk)
#endif

    // iteration k = 0 .. nk: producers compute the stem of item k into slab set k & 1 (from tile buffer k % 3), consumers run layer 1
    // of item k-1 from set (k-1) & 1; the tile of item k+2 is issued by every wave into buffer (k+2) % 3 = the one item k-1 was
    // read from in iteration k-1.  Both loops pass the same nk + 1 barriers.
    if (wave < 8) {
        // tap grid of a stem pixel in the tile: rho = 3 kh + ci (9 rows of IW floats), kw.  K slots (tap = rho*3 + kw, -1: zero filter):
        //   step 0, half 0: rows 0,1 and (2,0) (2,1); half 1: the same 3 rows down
        //   step 1, half 0: rows 6,7 and (2,2) (5,2); half 1: the same 2 rows down = row 8 (real), row 9, (4,2), (7,2) (zero filters)
        auto slotA = [](int e) constexpr { return e; };                                        // taps 0..7 = rows 0,1,(2,0),(2,1)
        auto slotC = [](int e) constexpr { return e < 6 ? 18 + e : (e == 6 ? 2 * 3 + 2 : 5 * 3 + 2); };
        auto goff = [](int t) constexpr { return (t / 3) * IW + t % 3; };                       // float offset of grid tap t
        auto kref = [](int t) constexpr { return ((t / 3) % 3) * 9 + (t / 9) * 3 + t % 3; };   // its index ci*9 + kh*3 + kw in w0
        vec8 wa[2];
        {
            const uint16_t* wr = s.w0 + c * 32;   // filters of output channel c
            uint16_t w0v[8], w1v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                w0v[e] = hh ? wr[kref(slotA(e) + 9)] : wr[kref(slotA(e))];
                const int td = slotC(e) + 6;      // half 1 of step 1: real only for grid row 8
                w1v[e] = hh ? (td / 3 == 8 ? wr[kref(td)] : (uint16_t)0) : wr[kref(slotC(e))];
            }
            wa[0] = __builtin_bit_cast(vec8, make_uint4(w0v[0] | (unsigned)w0v[1] << 16, w0v[2] | (unsigned)w0v[3] << 16,
                                                           w0v[4] | (unsigned)w0v[5] << 16, w0v[6] | (unsigned)w0v[7] << 16));
            wa[1] = __builtin_bit_cast(vec8, make_uint4(w1v[0] | (unsigned)w1v[1] << 16, w1v[2] | (unsigned)w1v[3] << 16,
                                                           w1v[4] | (unsigned)w1v[5] << 16, w1v[6] | (unsigned)w1v[7] << 16));
        }
        // stem scale / shift of this lane's 16 output channels 8q + 4hh + 2j2 (+1)
        f32x2 sc0[8], sh0[8];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) {
                const int ch = 8 * q + 4 * hh + 2 * j2;
                sc0[q * 2 + j2] = f32x2{s.scale0[ch], s.scale0[ch + 1]};
                sh0[q * 2 + j2] = f32x2{s.shift0[ch], s.shift0[ch + 1]};
            }
        const float slope0 = s.leaky0 ? 0.1f : 1.0f;
        // this wave's blocks: wave, wave + 8 and (waves 0-2) wave + 16; per lane: tile offsets of its pixel (K-step 0 / 1 bases), slab cell
        int pbase0[3], pbase1[3], pcell[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int P = (wave + 8 * j) * 32 + c;
            const int sy = P / SW, sx = P - sy * SW;
            const bool inside = P < S_PIX;
            const int px = inside ? sy * 3 * IW + sx + XOFF : XOFF;
            pbase0[j] = px + hh * 3 * IW;
            pbase1[j] = px + hh * 2 * IW;
            pcell[j] = (inside ? sy * ROWP + (sx & 1) * HALFW + (sx >> 1) : S_POS + (P - S_PIX)) * 16 + hh * 8;
        }
        for (int k = 0; k <= nk; ++k) {
            const bool issue2 = k + 2 < nk;
            Coord cC = cB;
            if (issue2) {
                cC = coord_of(k + 2);
                issue_image(cC, (k + 2) % NIB);
            }
            STEM2_TICK(0, 0)
            if (k < nk && !AY_DBGBIT(a, 16)) {
                const int y0 = cA.y0, x0 = cA.x0;
                const float* img = reinterpret_cast<const float*>(lds + (k % NIB) * IMG_BYTES);
                uint8_t* set = lds + OFF_STEM + (k & 1) * SET;
                // every stem pixel of the tile is a real one unless the tile touches the image border
                const bool interior = 2 * y0 - 1 >= 0 && 2 * y0 - 1 + SH <= s.H && 2 * x0 - 1 >= 0 && 2 * x0 - 1 + SW <= s.W;
                auto block = [&](int j) __attribute__((always_inline)) {
                    const float* p0 = img + pbase0[j];
                    const float* p1 = img + pbase1[j];
                    f32x2 v0[4], v1[4];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        v0[jj] = f32x2{p0[goff(slotA(2 * jj))], p0[goff(slotA(2 * jj + 1))]};
                        v1[jj] = f32x2{p1[goff(slotC(2 * jj))], p1[goff(slotC(2 * jj + 1))]};
                    }
                    const uint4 b0 = make_uint4(DT::pack2(v0[0]), DT::pack2(v0[1]), DT::pack2(v0[2]), DT::pack2(v0[3]));
                    const uint4 b1 = make_uint4(DT::pack2(v1[0]), DT::pack2(v1[1]), DT::pack2(v1[2]), DT::pack2(v1[3]));
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                    acc = DT::mfma32(wa[0], __builtin_bit_cast(vec8, b0), acc);
                    acc = DT::mfma32(wa[1], __builtin_bit_cast(vec8, b1), acc);
                    bool real = true;
                    if (!interior) {   // border tile: pixels outside the image are the zero padding of layer 1
                        const int P = (wave + 8 * j) * 32 + c;
                        const int sy = P / SW, sx = P - sy * SW;
                        const int gy = 2 * y0 - 1 + sy, gx = 2 * x0 - 1 + sx;
                        real = gy >= 0 && gy < s.H && gx >= 0 && gx < s.W;
                    }
                    uint8_t* dst = set + pcell[j];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x2 t0 = leaky2(f32x2{acc[4 * q], acc[4 * q + 1]} * sc0[2 * q] + sh0[2 * q], slope0);
                        const f32x2 t1 = leaky2(f32x2{acc[4 * q + 2], acc[4 * q + 3]} * sc0[2 * q + 1] + sh0[2 * q + 1], slope0);
                        uint2 o = make_uint2(DT::pack2(t0), DT::pack2(t1));
                        if (!real) o = make_uint2(0u, 0u);
                        *reinterpret_cast<uint2*>(dst + (q >> 1) * SLAB + (q & 1) * S_POSP * 16) = o;
                    }
                };
                block(0);
                block(1);
                if (wave + 16 < NBLK) block(2);
            }
            STEM2_TICK(0, 1)
            // the tile of item k+1 must have landed (issued in iteration k-1 or the prologue); younger: the DMA of tile k+2 if issued
            if (issue2)
                asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            cA = cB;
            cB = cC;
            STEM2_TICK(0, 2)
#ifdef AY_PHASE_CLOCK
            if (wave == 0) ++tk[6];
#endif
        }
    } else {
        // layer 1: 8 waves = 2 (32-channel halves) x 4 (output rows); filters [chunk][tap][half][64][8] straight from HBM into registers
        const int cwv = wave & 7;
        const int wn = cwv & 3, wm = cwv >> 2;
        vec8 wf[18];
#pragma unroll
        for (int n = 0; n < 18; ++n)
            wf[n] = *reinterpret_cast<const vec8*>(a.w + ((size_t)(n * 2 + hh) * BN + wm * 32 + c) * 16);
        // cell of input pixel (row 2*wn + kh, column 2*c + kw): (2*wn + kh)*ROWP + (kw & 1)*HALFW + c + (kw >> 1)
        const int pb = (hh * S_POSP + (wn * 2) * ROWP + c) * 16;
        Coord cP = cA;   // item k-1
        for (int k = 0; k <= nk; ++k) {
            const bool issue2 = k + 2 < nk;
            Coord cC = cB;
            if (issue2) {
                cC = coord_of(k + 2);
                issue_image(cC, (k + 2) % NIB);
            }
            bool stored = false;
            STEM2_TICK(8, 3)
            if (k >= 1 && !AY_DBGBIT(a, 32)) {
                const uint8_t* set = lds + OFF_STEM + ((k - 1) & 1) * SET;
                f32x16 acc1[MT][NT];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc1[0][0][r] = 0.f;
                vec8 fb[2][3];   // pixel fragments one filter row ahead of the MFMAs that use them
                auto load_row = [&](int n, int sl) __attribute__((always_inline)) {
                    const int ch = n / 3, kh = n % 3;
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw)
                        fb[sl][kw] = *reinterpret_cast<const vec8*>(set + ch * SLAB + pb + (kh * ROWP + (kw & 1) * HALFW + (kw >> 1)) * 16);
                };
                load_row(0, 0);
#pragma unroll
                for (int n = 0; n < 6; ++n) {
                    if (n + 1 < 6) load_row(n + 1, (n + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw)
                        acc1[0][0] = DT::mfma32(wf[n * 3 + kw], fb[n & 1][kw], acc1[0][0]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                STEM2_TICK(8, 4)
                ResRegs<MT, NT> rr;
                conv_epilogue<BN, MT, NT, TW, false, false, false, 2, false, false, DT>(a, acc1, rr, cP.b, 0, wm, wn, c, hh, cP.y0, cP.x0,
                                                                      reinterpret_cast<const float*>(lds + OFF_SS1));
                stored = (cP.y0 + wn) < a.hout;
            }
            STEM2_TICK(8, 5)
            // as for the producers, plus the two output stores of this wave's item (younger than both DMAs)
            if (issue2) {
                if (stored)
                    asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
                else
                    asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
            } else {
                if (stored)
                    asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
                else
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            cP = cA;
            cA = cB;
            cB = cC;
            STEM2_TICK(8, 7)
        }
    }
#ifdef AY_PHASE_CLOCK
    if ((wave == 0 || wave == 8) && lane == 0)
        for (int k = 0; k < 8; ++k) atomicAdd(&g_stem_ticks[k], tk[k]);
#endif
}

}  // namespace ay

namespace ay {
template <typename DT>
static int stem_s2_fused_fwd(const float* x_nchw, const void* stem_w_bf16, const float* scale0, const float* shift0, int leaky0,
                             const void* w1_packed, const float* scale1, const float* shift1, int leaky1, void* out_blocked,
                             int batch, int h, int w, ay_stream_t stream) {
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
    static const int v2_env = getenv("AY_STEM_V2") ? atoi(getenv("AY_STEM_V2")) : 1;   // 0: the serial-phase kernel (8x32 items)
    const bool v2 = v2_env && w % 4 == 0 && (reinterpret_cast<uintptr_t>(x_nchw) & 15) == 0 && 3LL * h * w * 4 < 0x7fffffffLL;
    a.tiles_x = (a.wout + 31) / 32;
    a.tiles_y = v2 ? (a.hout + 3) / 4 : (a.hout + 7) / 8;
    a.n_cgroups = 1;
    auto magic = [](unsigned d) { return d <= 1 ? 0xffffffffu : (unsigned)(0x100000000ULL / d); };
    s.m_tx = magic((unsigned)a.tiles_x);
    s.m_tpi = magic((unsigned)(a.tiles_x * a.tiles_y));
    a.leaky = leaky1;
    a.dbg = getenv("AY_DBG") ? atoi(getenv("AY_DBG")) : 0;   // read only by the instrumented build (AY_DBGBIT)
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
    if (v2)
        hipLaunchKernelGGL(stem_s2_fused_v2_kernel<DT>, grid, dim3(1024), 0, S(stream), s, (int)n_items);
    else
        hipLaunchKernelGGL(stem_s2_fused_kernel<DT>, grid, dim3(1024), 0, S(stream), s, (int)n_items);
    AY_CHECK_LAUNCH("stem_s2_fused_kernel");
#ifdef AY_PHASE_CLOCK
    if (getenv("AY_DBG") && (atoi(getenv("AY_DBG")) & 8)) {
        unsigned long long t[8] = {0}, z[8] = {0};
        (void)hipStreamSynchronize(S(stream));
        (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_stem_ticks), sizeof(t));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stem_ticks), z, sizeof(z));
        if (t[6] && v2)
            fprintf(stderr, "[ay stem v2] per iteration (us): producer issue %.2f, stem %.2f, wait %.2f | consumer issue %.2f, layer-1 %.2f, epilogue %.2f, wait %.2f\n",
                    t[0] * 0.01 / t[6], t[1] * 0.01 / t[6], t[2] * 0.01 / t[6], t[3] * 0.01 / t[6], t[4] * 0.01 / t[6], t[5] * 0.01 / t[6], t[7] * 0.01 / t[6]);
        else if (t[6])
            fprintf(stderr, "[ay stem] per item (us): dma issue %.2f, stem MFMA %.2f, barrier %.2f, layer-1 MFMA %.2f, epilogue %.2f, tail wait %.2f\n",
                    t[0] * 0.01 / t[6], t[1] * 0.01 / t[6], t[2] * 0.01 / t[6], t[3] * 0.01 / t[6], t[4] * 0.01 / t[6], t[5] * 0.01 / t[6]);
    }
#endif
    return AY_OK;
}
}  // namespace ay

extern "C" int ay_stem_s2_fused_fwd(const float* x_nchw, const void* stem_w_bf16, const float* scale0, const float* shift0, int leaky0,
                                    const void* w1_packed, const float* scale1, const float* shift1, int leaky1, void* out_blocked,
                                    int batch, int h, int w, ay_stream_t stream) {
    return ay::stem_s2_fused_fwd<ay::Bf16>(x_nchw, stem_w_bf16, scale0, shift0, leaky0, w1_packed, scale1, shift1, leaky1, out_blocked, batch, h, w, stream);
}
extern "C" int ay_stem_s2_fused_fwd_f16(const float* x_nchw, const void* stem_w_f16, const float* scale0, const float* shift0, int leaky0,
                                        const void* w1_packed, const float* scale1, const float* shift1, int leaky1, void* out_blocked,
                                        int batch, int h, int w, ay_stream_t stream) {
    return ay::stem_s2_fused_fwd<ay::F16>(x_nchw, stem_w_f16, scale0, shift0, leaky0, w1_packed, scale1, shift1, leaky1, out_blocked, batch, h, w, stream);
}
