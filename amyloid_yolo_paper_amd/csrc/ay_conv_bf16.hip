// Direct (im2col-free) convolution for gfx950: blocked-bf16 activations, MFMA 32x32x16 bf16, fp32
// accumulate, LDS-staged input halo tile + filter slab, fused BN-affine + LeakyReLU + residual epilogue.
//
// Replaces the cuDNN/ATen conv + BatchNorm2d + LeakyReLU (+ shortcut add) sequence the reference runs
// per block (models.py:26-45 executed at models.py:242-248).
//
// GEMM view (per image):  D[cout][pixel] = sum_{tap, cin} W[cout][tap][cin] * X[cin][pixel + tap]
//   A operand = filters  (rows = output channels), B operand = input pixels (cols), K = 16 input
//   channels of one filter tap per MFMA.  A workgroup owns BN output channels x (TH x TW) output pixels
//   of one image; per 16-channel input chunk it stages the (TH*s+2)x(TW*s+2) input halo tile ONCE and
//   re-reads it from LDS for all 9 taps (tap shift = immediate offset on the ds_read), so global->LDS
//   traffic for the input is ~1.3x instead of 9x.
//
// Layouts
//   activations  [B][C/16][H][W][16] bf16     ("c16 planes": one chunk of one tile row is contiguous)
//   weights      [Cin/16][tap][half][CoutPad][8] bf16   (half = input channels 0-7 / 8-15 of the chunk)
//   LDS pixels   [kstep][half][IN_PIX][8 bf16]  -> lane (pixel c, half h) reads 16 B, conflict-free
//   LDS filters  [kstep][tap][half][BN][8 bf16]
#include <stdlib.h>

#include <atomic>
#include <mutex>

#include "ay_conv_common.h"

namespace ay {

template <int KS, int STRIDE, int BN, int WM, int WN, int TH, int TW, int NK, bool OUT_F32, bool HAS_RES, typename DT = Bf16, bool DECODE = false>
__global__ void __launch_bounds__(256, 2) conv_bf16_kernel(ConvArgs a, DecodeArgs dd) {
    DT::enter();
    typedef typename DT::vec8 vec8;
    constexpr int PAD = (KS - 1) / 2;
    constexpr int KK2 = KS * KS;
    constexpr int NPIX = TH * TW;
    constexpr int NT = NPIX / (WN * 32);  // 32-pixel column blocks per wave
    constexpr int MT = BN / (WM * 32);    // 32-channel row blocks per wave
    constexpr int IN_H = (TH - 1) * STRIDE + KS;
    constexpr int IN_W = (TW - 1) * STRIDE + KS;
    constexpr int IN_PIX = IN_H * IN_W;
    constexpr int PIX_SLAB = 2 * IN_PIX * 16;
    constexpr int W_SLAB = KK2 * 2 * BN * 16;
    constexpr int W_BASE = NK * PIX_SLAB;
    constexpr int LDS_BYTES = NK * (PIX_SLAB + W_SLAB);
    constexpr int PXU_TOTAL = NK * 2 * IN_PIX;
    constexpr int NPXU = (PXU_TOTAL + 255) / 256;
    constexpr int WU_TOTAL = NK * KK2 * 2 * BN;
    constexpr int NWU = (WU_TOTAL + 255) / 256;
    static_assert(WM * WN == 4 && NT >= 1 && MT >= 1, "4 waves");
    static_assert(NT * WN * 32 == NPIX && MT * WM * 32 == BN, "tile split");
    static_assert(LDS_BYTES <= 80 * 1024, "two workgroups per CU");

    __shared__ __attribute__((aligned(16))) uint8_t lds[LDS_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int c = lane & 31, hh = lane >> 5;

    // workgroup -> (image, pixel tile, channel group); channel groups of one pixel tile are adjacent
    // so they hit the same staged input in L2.
    const int bid = blockIdx.x;
    const int cg = bid % a.n_cgroups;
    const int pt = bid / a.n_cgroups;
    const int tile_x = pt % a.tiles_x;
    const int tile_y = (pt / a.tiles_x) % a.tiles_y;
    const int b = pt / (a.tiles_x * a.tiles_y);
    const int y0 = tile_y * TH, x0 = tile_x * TW;

    const size_t in_plane = (size_t)a.hin * a.win * 32;
    const uint8_t* src_img = a.src + (size_t)b * (a.cin / 16) * in_plane;
    const int CP = a.cout_pad;
    const uint8_t* wbase = a.w + (size_t)cg * BN * 16;
    const size_t w_stage_stride = (size_t)NK * KK2 * 2 * CP * 16;

    // ---- per-thread staging map (fixed for the whole K loop) --------------------------------
    int px_off[NPXU];
#pragma unroll
    for (int i = 0; i < NPXU; ++i) {
        const int u = i * 256 + tid;
        int off = -1;
        if (u < PXU_TOTAL) {
            const int kk = u / (2 * IN_PIX);
            const int v = u % (2 * IN_PIX);
            const int P = v >> 1, h = v & 1;
            const int iy = y0 * STRIDE - PAD + P / IN_W;
            const int ix = x0 * STRIDE - PAD + P % IN_W;
            if (iy >= 0 && iy < a.hin && ix >= 0 && ix < a.win)
                off = (int)(kk * in_plane) + (iy * a.win + ix) * 32 + h * 16;
        }
        px_off[i] = off;
    }

    uint4 rpx[NPXU];
    uint4 rw[NWU];

    auto issue = [&](int s) {
        const uint8_t* sp = src_img + (size_t)s * NK * in_plane;
#pragma unroll
        for (int i = 0; i < NPXU; ++i) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (px_off[i] >= 0) v = *reinterpret_cast<const uint4*>(sp + px_off[i]);
            rpx[i] = v;
        }
        const uint8_t* wp = wbase + (size_t)s * w_stage_stride;
#pragma unroll
        for (int i = 0; i < NWU; ++i) {
            const int u = i * 256 + tid;
            if (u < WU_TOTAL) {
                const int r = u % BN;
                const int th = u / BN;  // (kk*KK2 + tap)*2 + half
                rw[i] = *reinterpret_cast<const uint4*>(wp + ((size_t)th * CP + r) * 16);
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < NPXU; ++i) {
            const int u = i * 256 + tid;
            if (u < PXU_TOTAL) {
                const int kk = u / (2 * IN_PIX);
                const int v = u % (2 * IN_PIX);
                const int P = v >> 1, h = v & 1;
                *reinterpret_cast<uint4*>(lds + kk * PIX_SLAB + (h * IN_PIX + P) * 16) = rpx[i];
            }
        }
#pragma unroll
        for (int i = 0; i < NWU; ++i) {
            const int u = i * 256 + tid;
            if (u < WU_TOTAL) *reinterpret_cast<uint4*>(lds + W_BASE + u * 16) = rw[i];
        }
    };

    // ---- fragment addresses -------------------------------------------------------------------
    int pb[NT];  // byte offset of this lane's pixel (tap 0,0) inside a pixel slab
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int p = (wn * NT + n) * 32 + c;
        const int ty = p / TW, tx = p % TW;
        pb[n] = (hh * IN_PIX + ty * STRIDE * IN_W + tx * STRIDE) * 16;
    }
    const int wa = W_BASE + (hh * BN + wm * MT * 32 + c) * 16;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const int nstages = a.cin / (16 * NK);
    issue(0);
    commit();
    __syncthreads();
    for (int s = 0; s < nstages; ++s) {
        const bool more = (s + 1 < nstages);
        if (more) issue(s + 1);
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
#pragma unroll
            for (int tap = 0; tap < KK2; ++tap) {
                const int kh = tap / KS, kw = tap % KS;
                vec8 af[MT], bfr[NT];
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    af[m] = *reinterpret_cast<const vec8*>(lds + wa + kk * W_SLAB + (tap * 2 * BN + m * 32) * 16);
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    bfr[n] = *reinterpret_cast<const vec8*>(lds + kk * PIX_SLAB + pb[n] + (kh * IN_W + kw) * 16);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        acc[m][n] = DT::mfma32(af[m], bfr[n], acc[m][n]);
            }
        }
        if (more) {
            __syncthreads();
            commit();
            __syncthreads();
        }
    }

    if constexpr (DECODE) {   // a detection head: the prediction rows instead of a head tensor
        head_decode_epilogue<BN, MT, NT, TW>(a, dd, acc, b, cg, wm, wn, c, hh, y0, x0);
    } else {
        ResRegs<MT, NT> rr;
        residual_prefetch<BN, MT, NT, TW, HAS_RES>(a, rr, b, cg, wm, wn, c, hh, y0, x0);
        conv_epilogue<BN, MT, NT, TW, OUT_F32, HAS_RES, false, 0, false, false, DT>(a, acc, rr, b, cg, wm, wn, c, hh, y0, x0);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Ring kernel: persistent, ALL staging by LDS-DMA into a ring of NBUF stage buffers, counted vmcnt, raw s_barrier.
//   * input pixels too go global -> LDS by DMA: lanes whose pixel falls outside the image (3x3 halo at the border,
//     ragged tiles) read from a 64-byte page of zeros instead (the DMA source address is per lane);
//   * stage g+NBUF-1 is issued before the MFMAs of stage g: NBUF-1 stages of DMA in flight per CU, which is what
//     the latency-bound layers (1x1, small Cin) need (ablation in DESIGN.md: staging alone ran at one stage/us);
//   * every wave issues the same number PW of 1-KiB DMA pieces per stage (the piece list is padded with dummy
//     pieces that copy zeros into a scratch KiB), so "stage g+1 has landed" is one constant `s_waitcnt vmcnt(PW)`.
__device__ __attribute__((aligned(64))) uint32_t g_zero_page[16];  // zero-initialised by the loader
// AY_DBG&8: phase clock of the ring kernel, summed over workgroups (wave 0): [0] stage loops, [1] epilogues, [2] items,
// [3] workgroups, [4] whole-kernel ticks per workgroup, [5] first stages, [6] slowest workgroup; 100 MHz ticks (s_memrealtime)
__device__ unsigned long long g_phase_ticks[8];
// dynamic item dealing: 64 rotating sets of {8 per-XCD item counters, exit counter}; zero at load, reset by the last workgroup
constexpr int DEAL_SETS = 64, DEAL_STREAMS = 16;
__device__ unsigned g_deal[DEAL_STREAMS * DEAL_SETS][16];

template <int KS, int STRIDE, int BN, int WM, int WN, int TH, int TW, int NK, int NBUF, bool HAS_RES, bool CAT = false, bool CANVAS = false,
          typename DT = Bf16, bool PAIR = false>
__global__ void __launch_bounds__(512, 2) conv_bf16_ring_kernel(ConvArgs a, int n_items) {
    DT::enter();
    typedef typename DT::vec8 vec8;
    // PAIR (2x2-window kernels only): the tile's BN "channels" are the two column-parity classes of BN / 2 real channels, side by side
    // (ay_conv_common.h: pair_block); the channel-group index then carries the ROW parity in its low bit
    static_assert(!PAIR || KS == 2, "class pairs: the parity-class data gradient");
    static_assert(!CANVAS || !CAT, "canvas tiling: single-source layers");
    // KS == 2: a 2x2 window with offsets {0, +1} (no padding on the low side) whose channel-group index carries an output pixel
    // parity class; see ConvArgs::w_class_stride
    constexpr bool UP2 = (KS == 2);
    static_assert(!UP2 || (STRIDE == 1 && !CAT && !CANVAS), "2x2-window kernels: stride 1, plain tiles");
    static_assert(!CAT || (KS == 1 && STRIDE == 1), "route + upsample folding exists for the 1x1 kernel");
    constexpr int PAD = (KS - 1) / 2;
    constexpr int KK2 = KS * KS;
    constexpr int NPIX = TH * TW;
    constexpr int NT = NPIX / (WN * 32);
    constexpr int MT = BN / (WM * 32);
    constexpr int IN_H = (TH - 1) * STRIDE + KS;
    constexpr int IN_W = (TW - 1) * STRIDE + KS;
    constexpr int IN_PIX = IN_H * IN_W;
    constexpr int PX_PIECES = (2 * IN_PIX + 63) / 64;       // 1-KiB DMA pieces per 16-channel pixel slab
    // Stride 2: the LDS image of a halo row keeps even and odd input columns apart -- [even cols, half 0 | even, half 1 | odd, half 0 |
    // odd, half 1], 2 IN_W cells per row -- so that the 32 pixels of a fragment read, which lie two columns apart, are CONSECUTIVE
    // 16-byte cells (conflict-free ds_read_b128, as the stride-1 image is); with the plain [half][pixel] image they sit 32 bytes apart
    // and every pixel-fragment read is a 2-way bank conflict (SQ_LDS_BANK_CONFLICT = 0.32 of SQ_LDS_IDX_ACTIVE on these layers,
    // profiles/r04_pmc_sq_v1.txt).  A DMA piece still covers both halves of its pixels, so the global side fetches the same lines
    // with the same number of requests.  -DAY_S2_DEINT=0 restores the plain image.
#ifndef AY_S2_DEINT
#define AY_S2_DEINT 1
#endif
    constexpr bool S2L = (STRIDE == 2 && KS == 3 && AY_S2_DEINT);
    constexpr int NEV = TW + 1;                             // even columns of a stride-2 halo row (IN_W = 2 TW + 1)
    constexpr int PIX_SLAB = PX_PIECES * 1024;
    constexpr int W_PIECES = KK2 * 2 * BN * 16 / 1024;      // per 16-channel filter slab
    constexpr int W_SLAB = W_PIECES * 1024;
    constexpr int W_BASE = NK * PIX_SLAB;
    constexpr int BUF_BYTES = NK * (PIX_SLAB + W_SLAB);
    constexpr int NPIECE = NK * (PX_PIECES + W_PIECES);     // + the piece(s) that carry this item's scale/shift
    constexpr int SSP = BN > 128 ? 2 : 1;                   // BN <= 128: one piece [scale | shift]; 256: a scale piece, a shift piece
    constexpr int SSR = SSP * 1024;                         // bytes of one scale/shift region
    // Piece i of a wave has the same KIND in every wave: i < PWP pixel pieces (global piece i*8 + wave of the pixel slabs), the
    // rest filter-side (filter pieces, then the scale/shift piece(s)); each list is padded to a multiple of 8 with dummy pieces
    // that copy zeros into a scratch KiB.  With the kind depending on the wave (one list i*8 + wave over all pieces) every DMA
    // issue went through ~6 scalar branches; now the unrolled issue code is straight-line with a few scalar selects.
    constexpr int PWP = (NK * PX_PIECES + 7) / 8;
    constexpr int PWW = (NK * W_PIECES + SSP + 7) / 8;
    constexpr int PW = PWP + PWW;                           // pieces per wave per stage (dummy-padded)
    constexpr int DUMMY_BASE = NBUF * BUF_BYTES;
    constexpr int SS_BASE = DUMMY_BASE + 1024;              // 4 x 1 KiB [scale 128][shift 128], by item index & 3 (the loader
                                                            // runs at most NBUF-1 <= 2 items ahead of the epilogue)
    constexpr int MBOX_BASE = NBUF * BUF_BYTES + 1024 + 4 * SSR;  // 8 ints: item ids by sequence number & 7
    constexpr int LDS_BYTES = MBOX_BASE + 64;
    static_assert(WM * WN == 8 && NT >= 1 && MT >= 1, "8 waves");
    static_assert(NT * WN * 32 == NPIX && MT * WM * 32 == BN, "tile split");
    static_assert((KK2 * 2 * BN * 16) % 1024 == 0, "filter slab is whole DMA pieces");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS ring");
    static_assert(NBUF == 2 || NBUF == 3, "ring depth");

    __shared__ __attribute__((aligned(16))) uint8_t lds[LDS_BYTES];
    // base priority 2, MFMA groups 3: a latency-bound neighbour on the CU (the merge-NMS wavefront of the previous batch runs at
    // priority 0) only gets the issue slots these waves leave free
    __builtin_amdgcn_s_setprio(2);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int c = lane & 31, hh = lane >> 5;

    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int per_xcd = (n_items + 7) >> 3;
    const int first = xcd * per_xcd;
    const int last = min(first + per_xcd, n_items);
    int item = first + slot;
    // ---- item dealing --------------------------------------------------------------------------------------------
    // A workgroup's first item is static; the following ones come from its XCD's atomic counter (a.deal), so that a slow
    // CU (HBM channel luck, a neighbour kernel on the CU) simply takes fewer items instead of setting the kernel time.
    // Item ids travel through an 8-entry LDS mailbox indexed by sequence number: thread 0 fetches id[c+D] at the start of the
    // epilogue of item c (the atomic's latency hides under it) and posts it at its end; the stage barriers publish it long
    // before the loader (at most 2 item boundaries ahead) or the MFMA side need it.  Without counters the same mailbox
    // carries the static ids.
    constexpr bool MAILBOX = true;  // (false: plain strided dealing without any mailbox traffic)
    const bool dyn = MAILBOX && a.deal != nullptr;
    auto leave = [&]() __attribute__((always_inline)) {
        if (dyn && tid == 0) {
            const unsigned d = atomicAdd(a.deal + 8, 1u);
            if (d == gridDim.x - 1) {  // last workgroup out: hand the counter set back zeroed
#pragma unroll
                for (int i = 0; i < 9; ++i) atomicExch(a.deal + i, 0u);
            }
        }
    };
    if (item >= last) {
        leave();
        return;
    }
    // explicit LDS address space: through a generic pointer these volatile accesses become flat_load/flat_store, which count on
    // vmcnt as well, and hipcc then waits vmcnt(0) -- draining the DMA ring -- at every mailbox access
    typedef volatile __attribute__((address_space(3))) int lds_vint;
    lds_vint* mbox = (lds_vint*)(__attribute__((address_space(3))) int*)(lds + MBOX_BASE);
    const int D = (a.cin / (16 * NK)) >= 2 ? 3 : 5;  // fetch-ahead distance in items
    auto fetch_id = [&](int prev) __attribute__((always_inline)) -> int {  // thread 0 only
        if (prev >= last) return last;
        if (!dyn) return prev + slots;
        // whatever the counter holds, the id stays inside this XCD's range or reads as "no more items"
        const unsigned n = atomicAdd(a.deal + xcd, 1u);
        return n < (unsigned)(last - first) ? first + D * slots + (int)n : last;
    };
    if (MAILBOX && tid < D) mbox[tid] = min(item + tid * slots, last);  // the first D items of a workgroup are static (no atomics, no
                                                             // wait in the prologue); id[c+D] is posted by the epilogue of item c
    if constexpr (MAILBOX) __syncthreads();  // ids 0..D-1 posted (a single-stage item makes the loader ask for id 1 already in the prologue below)
    int seq_l = 0, seq_c = 0;  // sequence numbers of the loader's / the MFMA side's current item

    const size_t in_plane = (size_t)a.hin * a.win * 32;
    const int CP = a.cout_pad;
    const size_t w_stage_stride = (size_t)NK * KK2 * 2 * CP * 16;
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const uint8_t* zero_page = reinterpret_cast<const uint8_t*>(g_zero_page);
    const unsigned lds_base = lds_addr_of(lds);

    // ---- loader: piece i of this wave is global piece q = i*8 + wave ------------------------------------------
    // kind: pixel piece (kk, j) | filter piece (kk, j) | dummy.  Per lane: byte offset from the item's base, or -1.
    int src_off[PW];
    // CAT (models.py:244-245 route of [upsampled x2 | direct] folded into this 1x1): stages below c1 channels read `src1`
    // at half resolution (pixel (y>>1, x>>1)), the rest `src`; c1 is a multiple of the stage's NK*16 channels
    int src_off1[CAT ? PW : 1];
    const uint8_t* ld_src1 = nullptr;
    const size_t in_plane1 = (size_t)(a.hin >> 1) * (a.win >> 1) * 32;
    const int s1_stages = CAT ? a.c1 / (16 * NK) : 0;
    const uint8_t* ld_src = nullptr;
    const uint8_t* ld_w = nullptr;
    int ld_item = item, ld_s = 0, ld_par = 0;
    bool ld_done = false;
    const float* ld_ss = nullptr;  // per-lane source of the scale/shift piece (nullptr: zero page)
    auto setup_loader = [&](int it) __attribute__((always_inline)) {
        int cg = it % a.n_cgroups;
        const int pt = it / a.n_cgroups;
        const int b = pt / tiles_per_img;
        const int y0 = ((pt / a.tiles_x) % a.tiles_y) * TH, x0 = (pt % a.tiles_x) * TW;
        ld_src = a.src + (size_t)b * ((a.cin - (CAT ? a.c1 : 0)) / 16) * in_plane;
        if constexpr (CAT) ld_src1 = a.src1 + (size_t)b * (a.c1 / 16) * in_plane1;
        const uint8_t* wbase = a.w;
        if constexpr (PAIR) {   // classes (2 py, 2 py + 1): two consecutive filter images
            wbase += (size_t)((cg & 1) * 2) * a.w_class_stride;
            cg >>= 1;
        } else if constexpr (UP2) {
            wbase += (size_t)(cg & 3) * a.w_class_stride;
            cg >>= 2;
        }
        constexpr int BNR = PAIR ? BN / 2 : BN;   // real channels of the group
        ld_w = wbase + (size_t)cg * BNR * 16;
        // lanes 0..BN/4-1 fetch 4 scales each, lanes 32..32+BN/4-1 the shifts: LDS image [scale | pad to 128][shift]
        // (PAIR: tile channel t is real channel t % BNR of the group)
        if constexpr (SSP == 1)
            ld_ss = (lane & 31) < BN / 4 ? ((lane < 32 ? a.scale : a.shift) + (size_t)cg * BNR + ((lane & 31) * 4) % BNR) : nullptr;
        else  // 256 channels: lane l carries scales (piece NPIECE) / shifts (piece NPIECE + 1) 4l..4l+3
            ld_ss = a.scale + (size_t)cg * BNR + (lane * 4) % BNR;
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            int off = -1, off1 = -1;
            if (i < PWP) {
                const int q = i * 8 + wave;  // pixel piece; q >= NK * PX_PIECES: padding
                const int kk = q / PX_PIECES, j = q % PX_PIECES;
                const int u = j * 64 + lane;           // unit inside the slab, LDS order [half][IN_PIX] (S2L: see above)
                int h = u / IN_PIX, P = u % IN_PIX;
                if constexpr (S2L) {
                    const int r = u / (2 * IN_W), v = u % (2 * IN_W);
                    const bool ev = v < 2 * NEV;
                    const int w2 = ev ? v : v - 2 * NEV;
                    const int per = ev ? NEV : TW;
                    h = r < IN_H ? w2 / per : 2;       // units past the image: padding of the piece list
                    P = r * IN_W + 2 * (w2 % per) + (ev ? 0 : 1);
                }
                int iy = y0 * STRIDE - PAD + P / IN_W;
                int ix = x0 * STRIDE - PAD + P % IN_W;
                bool inside = iy >= 0 && iy < a.hin && ix >= 0 && ix < a.win;
                int img = 0;  // canvas mode: this lane's image; its planes lie img * (cin/16) planes further on
                if constexpr (CANVAS) inside = canvas_px_in<STRIDE>(a, iy, ix, img, iy, ix);
                if (q < NK * PX_PIECES && h < 2 && inside) {
                    off = (int)(kk * in_plane) + (iy * a.win + ix) * 32 + h * 16 + img * (a.cin / 16) * (int)in_plane;
                    off1 = (int)(kk * in_plane1) + ((iy >> 1) * (a.win >> 1) + (ix >> 1)) * 32 + h * 16;
                }
            } else {
                const int qq = (i - PWP) * 8 + wave;   // filter piece; qq >= NK * W_PIECES: scale/shift or padding
                const int u = qq * 64 + lane;          // unit inside the stage's filter image [kk][tap][half][BN]
                const int r = u % BN, th = u / BN;
                if (qq < NK * W_PIECES) {
                    if constexpr (PAIR)   // rows BN/2.. of the tile: the second class's image, w_class_stride further on
                        off = (th * CP + r % (BN / 2)) * 16 + (r / (BN / 2)) * (int)a.w_class_stride;
                    else
                        off = (th * CP + r) * 16;
                }
            }
            src_off[i] = off;
            if constexpr (CAT) src_off1[i] = off1;
        }
    };
    auto issue_piece = [&](int i, int buf) __attribute__((always_inline)) {  // DMA piece i of the loader's current stage into ring slot `buf`
        const uint8_t* zp = zero_page + (lane & 3) * 16;
        const uint8_t* g;
        int dst;
        if (i < PWP) {  // `i` is a constant after unrolling: one kind per call site, the rest is selects
            const bool from1 = CAT && ld_s < s1_stages;  // wave-uniform
            const uint8_t* sp = from1 ? ld_src1 + (size_t)ld_s * NK * in_plane1 : ld_src + (size_t)(ld_s - s1_stages) * NK * in_plane;
            const int q = i * 8 + wave;  // wave-uniform
            const int off0 = src_off[i], off1 = src_off1[CAT ? i : 0];  // both read by value: a select between the arrays
            const int off = from1 ? off1 : off0;                        // puts them on the stack (seen: 64 B of scratch)
            g = off >= 0 ? sp + off : zp;
            dst = q < NK * PX_PIECES ? buf * BUF_BYTES + (q / PX_PIECES) * PIX_SLAB + (q % PX_PIECES) * 1024 : DUMMY_BASE;
        } else {
            const uint8_t* wp = ld_w + (size_t)ld_s * w_stage_stride;
            const int qq = (i - PWP) * 8 + wave;  // wave-uniform
            const bool is_w = qq < NK * W_PIECES;
            const bool is_ss = qq >= NK * W_PIECES && qq < NK * W_PIECES + SSP;
            // scale/shift piece(s): lane pointer ld_ss (nullptr: lane carries nothing); the shift array at the same lane offset
            const float* ssp = (SSP == 2 && qq == NK * W_PIECES + 1) ? ld_ss + (a.shift - a.scale) : ld_ss;
            const uint8_t* gss = (is_ss && ld_ss != nullptr) ? reinterpret_cast<const uint8_t*>(ssp) : zp;
            g = is_w ? wp + src_off[i] : gss;
            dst = is_w ? buf * BUF_BYTES + W_BASE + qq * 1024
                       : (is_ss ? SS_BASE + ld_par * SSR + (qq - NK * W_PIECES) * 1024 : DUMMY_BASE);
        }
        dma16(g, lds_base + __builtin_amdgcn_readfirstlane(dst));
    };
    auto advance_loader = [&]() __attribute__((always_inline)) {
        if (++ld_s == a.cin / (16 * NK)) {
            ld_s = 0;
            ld_par = (ld_par + 1) & 3;
            ++seq_l;
            if constexpr (MAILBOX)
                ld_item = __builtin_amdgcn_readfirstlane(mbox[seq_l & 7]);
            else
                ld_item += slots;
            if (ld_item < last)
                setup_loader(ld_item);
            else
                ld_done = true;
        }
    };
    auto issue_stage = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PW; ++i) issue_piece(i, buf);
        advance_loader();
    };

    int pb[NT], pbo[S2L ? NT : 1];   // pbo (S2L): the odd-column cells (kw = 1)
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int p = (wn * NT + n) * 32 + c;
        const int ty = p / TW, tx = p % TW;
        pb[n] = (hh * IN_PIX + ty * STRIDE * IN_W + tx * STRIDE) * 16;
        if constexpr (S2L) {
            pb[n] = (ty * 2 * (2 * IN_W) + hh * NEV + tx) * 16;
            pbo[n] = (ty * 2 * (2 * IN_W) + 2 * NEV + hh * TW + tx) * 16;
        }
    }
    const int wa = W_BASE + (hh * BN + wm * MT * 32 + c) * 16;
    const int nstages = a.cin / (16 * NK);

    // de-phase the workgroups: identical items on every CU otherwise put all epilogues (the HBM-heavy phase) at the
    // same instants and leave HBM idle during the MFMA phases
    for (int k = 0; k < (slot & 3) * a.stagger; ++k) __builtin_amdgcn_s_sleep(127);

    // ---- prologue: NBUF-1 stages in flight, stage 0 landed -----------------------------------------------------
    setup_loader(item);
    issue_stage(0);
    if constexpr (NBUF == 3) {
        if (!ld_done) {
            issue_stage(1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    int cur = 0;  // ring slot of the stage the MFMAs read
    int par = 0;  // scale/shift region of the item the MFMAs work on
    // phase clock (instrumented build only, see AY_DBGBIT): wave-uniform tick counters
    const bool clk = AY_DBGBIT(a, 8) && wave == 0;
    unsigned long long tk_stage = 0, tk_epi = 0, tk_items = 0, tk0 = 0, tk_begin = 0, tk_s0 = 0;
    if (clk) tk_begin = wall_clock64();
    while (true) {
        if (clk) tk0 = wall_clock64();
        int cg = item % a.n_cgroups;
        const int cls = PAIR ? (cg & 1) * 2 : UP2 ? (cg & 3) : 0;
        if constexpr (PAIR)
            cg >>= 1;
        else if constexpr (UP2)
            cg >>= 2;
        const int pt = item / a.n_cgroups;
        const int b = pt / tiles_per_img;
        const int y0 = ((pt / a.tiles_x) % a.tiles_y) * TH, x0 = (pt % a.tiles_x) * TW;
        const int next_item = MAILBOX ? __builtin_amdgcn_readfirstlane(mbox[(seq_c + 1) & 7]) : item + slots;
        const bool has_next = next_item < last;

        f32x16 acc[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
        ResRegs<MT, NT> rr;

        for (int s = 0; s < nstages; ++s) {
            const bool last_stage = (s + 1 == nstages);
            // stage g+NBUF-1 -> the slot that was read during stage g-1 (every wave passed the barrier since)
            // the PW DMA pieces of stage g+NBUF-1 are issued one by one behind the MFMA groups of this stage (an LDS-DMA
            // issue costs the wave 60-180 cycles; behind 4-8 queued MFMAs it is hidden, in a burst at the stage start
            // both waves of a SIMD pay it at the same time)
            const bool issued = !ld_done && !AY_DBGBIT(a, 1);
            int slot_ld = cur + (NBUF - 1);
            if (slot_ld >= NBUF) slot_ld -= NBUF;
            constexpr bool EARLY_RES = (MT * NT <= 4);  // 32 VGPRs of residual; larger wave tiles load it in the epilogue
            if (EARLY_RES && last_stage) residual_prefetch<BN, MT, NT, TW, HAS_RES, CANVAS, UP2, PAIR>(a, rr, b, cg, wm, wn, c, hh, y0, x0, cls);

            const uint8_t* L = lds + cur * BUF_BYTES;
            constexpr int NSTEP = NK * KK2;
            vec8 af[2][MT], bfr[2][NT];
            auto load_frags = [&](int t, vec8 (&fa)[MT], vec8 (&fb)[NT]) __attribute__((always_inline)) {
                const int kk = t / KK2, tap = t % KK2;
                const int kh = tap / KS, kw = tap % KS;
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    fa[m] = *reinterpret_cast<const vec8*>(L + wa + kk * W_SLAB + (tap * 2 * BN + m * 32) * 16);
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    if constexpr (S2L)   // row kh of the halo: 2 IN_W cells further on; kw = 0 / 2: even cells tx / tx + 1, kw = 1: odd cell tx
                        fb[n] = *reinterpret_cast<const vec8*>(L + kk * PIX_SLAB + (kw == 1 ? pbo[n] : pb[n]) + (kh * 2 * IN_W + (kw == 2 ? 1 : 0)) * 16);
                    else
                        fb[n] = *reinterpret_cast<const vec8*>(L + kk * PIX_SLAB + pb[n] + (kh * IN_W + kw) * 16);
                }
            };
            if (!AY_DBGBIT(a, 2)) load_frags(0, af[0], bfr[0]);
#pragma unroll
            for (int t = 0; t < NSTEP; ++t) {
                if (AY_DBGBIT(a, 2)) break;
                if (t + 1 < NSTEP && !AY_DBGBIT(a, 64)) load_frags(t + 1, af[(t + 1) & 1], bfr[(t + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                // (unequal priorities for the two waves of a SIMD were measured slower: 21.4 vs 20.0 us compute-only)
                __builtin_amdgcn_s_setprio(3);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        acc[m][n] = DT::mfma32(af[t & 1][m], bfr[t & 1][n], acc[m][n]);
                __builtin_amdgcn_s_setprio(2);
                __builtin_amdgcn_sched_barrier(0);
                if (issued) {
#pragma unroll
                    for (int i = 0; i < PW; ++i)  // constant trip count: src_off[] must stay in registers
                        if (i >= t * PW / NSTEP && i < (t + 1) * PW / NSTEP) issue_piece(i, slot_ld);
                }
            }
            if (issued) advance_loader();
            // stage g+1 must have landed before anyone reads it: everything but the stage(s) issued after it
            if (!(last_stage && !has_next)) {
                if constexpr (NBUF == 3) {
                    if (issued)
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
                    else
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            if (++cur == NBUF) cur = 0;
            if (clk && s == 0) tk_s0 += wall_clock64() - tk0;
        }
        if (clk) {
            const unsigned long long t = wall_clock64();
            tk_stage += t - tk0;
            tk0 = t;
        }
        int fetched = last;
        if (MAILBOX && tid == 0) fetched = fetch_id(mbox[(seq_c + D - 1) & 7]);  // id[c+D]; consumed after the epilogue
        conv_epilogue<BN, MT, NT, TW, false, HAS_RES, (MT * NT > 4), 1, CANVAS, UP2, DT, PAIR>(a, acc, rr, b, cg, wm, wn, c, hh, y0, x0,
                                                                            reinterpret_cast<const float*>(lds + SS_BASE + par * SSR), cls);
        if (MAILBOX && tid == 0) {
            mbox[(seq_c + D) & 7] = fetched;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (clk) {
            tk_epi += wall_clock64() - tk0;
            ++tk_items;
        }
        if (!has_next) break;
        item = next_item;
        ++seq_c;
        par = (par + 1) & 3;
    }
    leave();
    if (clk && lane == 0) {
        atomicAdd(&g_phase_ticks[0], tk_stage);
        atomicAdd(&g_phase_ticks[1], tk_epi);
        atomicAdd(&g_phase_ticks[2], tk_items);
        atomicAdd(&g_phase_ticks[3], 1ull);
        atomicAdd(&g_phase_ticks[4], wall_clock64() - tk_begin);
        atomicAdd(&g_phase_ticks[5], tk_s0);
        atomicMax(&g_phase_ticks[6], wall_clock64() - tk_begin);
    }
}

template <int KS, int STRIDE, int BN, int TH, int TW, int NK>
constexpr int ring_depth() {
    constexpr int in_pix = ((TH - 1) * STRIDE + KS) * ((TW - 1) * STRIDE + KS);
    constexpr int buf = NK * (((2 * in_pix + 63) / 64) * 1024 + KS * KS * 2 * BN * 16);
    return (3 * buf + 1024 + 4 * (BN > 128 ? 2048 : 1024) + 64 <= 160 * 1024) ? 3 : 2;
}

static int current_device() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return dev < 0 || dev >= 64 ? 0 : dev;
}
int conv_num_cus() {
    static std::atomic<int> cus[64];  // per device
    const int dev = current_device();
    int n = cus[dev].load(std::memory_order_relaxed);
    if (!n) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n < 8) n = 256;
        if (getenv("AY_CUS")) n = atoi(getenv("AY_CUS"));  // timing experiments only
        n -= n % 8;
        cus[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

// Counter set of the next ring-kernel launch on `st` (nullptr: static dealing).  The sets of one stream are used in rotation
// by launches that the stream serialises -- the last workgroup of a launch hands its set back zeroed before the next launch
// on that stream starts -- so a set must never be shared by two streams, whose launches may overlap: every (device, stream)
// pair owns DEAL_SETS sets of its own, up to DEAL_STREAMS streams per device; further streams deal statically.  A set pointer
// baked into a captured graph node stays valid under the same rule: replay the graph on the stream it was captured on, or on
// any stream as long as nothing else on the CAPTURE stream runs concurrently.
unsigned* next_deal_set(hipStream_t st) {
    static const int dynamic = getenv("AY_DYNAMIC") ? atoi(getenv("AY_DYNAMIC")) : 1;
    if (!dynamic) return nullptr;
    struct PerDevice {
        unsigned* base = nullptr;
        hipStream_t streams[DEAL_STREAMS] = {};
        unsigned seq[DEAL_STREAMS] = {};
        int n = 0;
    };
    static std::mutex mu;
    static PerDevice devs[64];
    std::lock_guard<std::mutex> lock(mu);
    PerDevice& pd = devs[current_device()];
    if (!pd.base) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_deal)) != hipSuccess) return nullptr;  // this device's copy of the symbol
        pd.base = (unsigned*)p;
    }
    int k = 0;
    while (k < pd.n && pd.streams[k] != st) ++k;
    if (k == pd.n) {
        if (pd.n == DEAL_STREAMS) return nullptr;
        pd.streams[pd.n++] = st;
    }
    return pd.base + ((size_t)k * DEAL_SETS + (pd.seq[k]++ % DEAL_SETS)) * 16;
}

// canvas tiling (ConvArgs::canvas_gx) applies to stride-1 same-size layers that save tiles that way and whose tensors stay below
// 2 GiB (per-lane image offsets are 32-bit); AY_CANVAS=0 turns it off.
// Returns the number of images per canvas row (0: tile image by image) and the canvas' tile grid.
static int canvas_plan(const ay_conv_desc* d, int th, int tw, int* tiles_x, int* tiles_y) {
    static const int on = getenv("AY_CANVAS") ? atoi(getenv("AY_CANVAS")) : 1;
    const long long px = (long long)d->hout * d->wout;
    if (!on || d->out_f32 || d->batch < 2) return 0;
    if (d->stride == 1 ? (d->hin != d->hout || d->win != d->wout) : (d->hin != 2 * d->hout || d->win != 2 * d->wout)) return 0;
    if ((long long)d->hin * d->win * d->batch * d->cin * 2 >= (1ll << 31) || px * d->batch * d->cout_pad * 2 >= (1ll << 31)) return 0;
    const long long image_tiles = (long long)d->batch * ((d->hout + th - 1) / th) * ((d->wout + tw - 1) / tw);
    long long best = image_tiles;
    int best_gx = 0;
    for (int gx = 1; gx <= d->batch && gx <= 64; ++gx) {  // images per canvas row: the one that needs the fewest tiles
        const int rows = (d->batch + gx - 1) / gx;
        const long long tx = ((long long)gx * (d->wout + 1) + tw - 1) / tw, ty = ((long long)rows * (d->hout + 1) + th - 1) / th;
        if (tx * ty < best) {
            best = tx * ty;
            best_gx = gx;
            *tiles_x = (int)tx;
            *tiles_y = (int)ty;
        }
    }
    return best * 10 <= image_tiles * 9 ? best_gx : 0;  // worth it from a tenth fewer tiles
}

static void fill_args(ConvArgs& a, const ay_conv_desc* d, const void* src, const void* w, const float* scale, const float* shift,
                      const void* residual, void* out, int TH, int TW, int BN) {
    a.src = (const uint8_t*)src;
    a.w = (const uint8_t*)w;
    a.scale = scale;
    a.shift = shift;
    a.residual = (const uint8_t*)residual;
    a.out = (uint8_t*)out;
    a.batch = d->batch;
    a.cin = d->cin;
    a.cout_pad = d->cout_pad;
    a.hin = d->hin;
    a.win = d->win;
    a.hout = d->hout;
    a.wout = d->wout;
    a.tiles_x = (d->wout + TW - 1) / TW;
    a.tiles_y = (d->hout + TH - 1) / TH;
    a.n_cgroups = d->cout_pad / BN;
    a.leaky = d->leaky;
    a.dbg = 0;
    a.stagger = 0;
    a.deal = nullptr;
    a.src1 = nullptr;
    a.c1 = 0;
    a.canvas_gx = 0;
    a.w_class_stride = 0;
}

// RING = false: the 4-wave register-staged kernel (fp32-output heads, cout_pad not a multiple of 64); RING = true: the persistent
// all-DMA ring kernel
template <int KS, int STRIDE, int BN, int WM, int WN, int TH, int TW, int NK, bool OUT_F32, bool RING = false, typename DT = Bf16>
static int launch(const ay_conv_desc* d, const void* src, const void* w, const float* scale, const float* shift,
                  const void* residual, void* out, hipStream_t st) {
    ConvArgs a;
    fill_args(a, d, src, w, scale, shift, residual, out, TH, TW, BN);
    static const int dbg = getenv("AY_DBG") ? atoi(getenv("AY_DBG")) : 0;
    a.dbg = dbg;
    static const int stagger = getenv("AY_STAGGER") ? atoi(getenv("AY_STAGGER")) : 0;
    a.stagger = stagger;
    long long nblk = (long long)a.tiles_x * a.tiles_y * d->batch * a.n_cgroups;
    int ctx = 0, cty = 0;
    if (RING && (a.canvas_gx = canvas_plan(d, TH, TW, &ctx, &cty)) > 0) {
        // images that leave much of their tiles empty: tile a canvas of gx images per row with one-pixel gutters instead
        a.tiles_x = ctx;
        a.tiles_y = cty;
        nblk = (long long)ctx * cty * a.n_cgroups;  // the kernel's item decode then yields image 0 and (y0, x0) on the canvas
    } else {
        a.canvas_gx = 0;
    }
    if (nblk <= 0 || nblk > 0x7fffffffLL) {
        set_error("conv grid out of range (%lld)", nblk);
        return AY_ERR_ARG;
    }
    if constexpr (RING) {
        static_assert(!OUT_F32, "the ring kernel writes bf16");
        // (the 1x1 kernels once measured 5-20 % slower with dynamic dealing: that was the flat-addressed mailbox draining the DMA
        // ring, not the counter fetch; with the LDS-typed mailbox they gain slightly, AY_DYN1=0 turns it off for them)
        static const int dyn1 = getenv("AY_DYN1") ? atoi(getenv("AY_DYN1")) : 1;
        if (KS == 3 || dyn1) a.deal = next_deal_set(st);
        const int per_xcd = (int)((nblk + 7) / 8);
        const int cu_slots = conv_num_cus() / 8;
        dim3 pgrid((unsigned)(8 * (per_xcd < cu_slots ? per_xcd : cu_slots))), block(512);
        constexpr int NBUF = ring_depth<KS, STRIDE, BN, TH, TW, NK>();
        if (a.canvas_gx && residual)
            hipLaunchKernelGGL((conv_bf16_ring_kernel<KS, STRIDE, BN, WM, WN, TH, TW, NK, NBUF, true, false, true, DT>), pgrid, block, 0, st, a, (int)nblk);
        else if (a.canvas_gx)
            hipLaunchKernelGGL((conv_bf16_ring_kernel<KS, STRIDE, BN, WM, WN, TH, TW, NK, NBUF, false, false, true, DT>), pgrid, block, 0, st, a, (int)nblk);
        else if (residual)
            hipLaunchKernelGGL((conv_bf16_ring_kernel<KS, STRIDE, BN, WM, WN, TH, TW, NK, NBUF, true, false, false, DT>), pgrid, block, 0, st, a, (int)nblk);
        else
            hipLaunchKernelGGL((conv_bf16_ring_kernel<KS, STRIDE, BN, WM, WN, TH, TW, NK, NBUF, false, false, false, DT>), pgrid, block, 0, st, a, (int)nblk);
    } else {
        dim3 grid((unsigned)nblk), block(256);
        if constexpr (OUT_F32) {
            hipLaunchKernelGGL((conv_bf16_kernel<KS, STRIDE, BN, WM, WN, TH, TW, NK, true, false, DT>), grid, block, 0, st, a, DecodeArgs{});
        } else {
            if (residual)
                hipLaunchKernelGGL((conv_bf16_kernel<KS, STRIDE, BN, WM, WN, TH, TW, NK, false, true, DT>), grid, block, 0, st, a, DecodeArgs{});
            else
                hipLaunchKernelGGL((conv_bf16_kernel<KS, STRIDE, BN, WM, WN, TH, TW, NK, false, false, DT>), grid, block, 0, st, a, DecodeArgs{});
        }
    }
    AY_CHECK_LAUNCH("conv_bf16_kernel");
#ifdef AY_PHASE_CLOCK
    if (RING && (dbg & 8)) {  // timing experiments only: synchronous phase report per launch
        unsigned long long t[8] = {0};
        (void)hipStreamSynchronize(st);
        (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_phase_ticks), sizeof(t));
        if (t[3])
            fprintf(stderr, "[ay phase] k%d s%d BN%d tile%dx%d cin%d cout%d h%d res%d: items/wg %.1f  per item: stages %.2f us (first stage %.2f of %d), epilogue %.2f us; wg total %.1f us (slowest %.1f)\n",
                    KS, STRIDE, BN, TH, TW, d->cin, d->cout, d->hout, residual ? 1 : 0, (double)t[2] / t[3], t[0] * 0.01 / t[2],
                    t[5] * 0.01 / t[2], d->cin / (16 * NK), t[1] * 0.01 / t[2], t[4] * 0.01 / t[3], t[6] * 0.01);
        unsigned long long z[8] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_ticks), z, sizeof(z));
    }
#endif
    return AY_OK;
}

// 1x1 ring kernel launched directly (no residual): CAT = over the route [nearest-x2-upsampled src1 | src2]
// (models.py:86-96,244-245) without materialising it; BN = 256 = all of a 256-channel group per workgroup (twice the MFMAs per
// stage barrier of the 128-channel tile, input pixels read once per 256 instead of per 128 output channels)
template <int BN, int WM, int WN, bool CAT, typename DT = Bf16>
static int launch_ring1x1(const ay_conv_desc* d, const void* src1, int c1, const void* src2, const void* w, const float* scale,
                          const float* shift, void* out, hipStream_t st) {
    constexpr int TH = 8, TW = 32, NK = 4;
    ConvArgs a;
    fill_args(a, d, src2, w, scale, shift, nullptr, out, TH, TW, BN);
    a.src1 = (const uint8_t*)src1;
    a.c1 = c1;
    long long nblk = (long long)a.tiles_x * a.tiles_y * d->batch * a.n_cgroups;
    a.canvas_gx = 0;
    int ctx = 0, cty = 0;
    if (!CAT && (a.canvas_gx = canvas_plan(d, TH, TW, &ctx, &cty)) > 0) {
        a.tiles_x = ctx;
        a.tiles_y = cty;
        nblk = (long long)ctx * cty * a.n_cgroups;
    } else {
        a.canvas_gx = 0;
    }
    if (nblk <= 0 || nblk > 0x7fffffffLL) {
        set_error("conv grid out of range (%lld)", nblk);
        return AY_ERR_ARG;
    }
    const int per_xcd = (int)((nblk + 7) / 8);
    const int cu_slots = conv_num_cus() / 8;
    dim3 pgrid((unsigned)(8 * (per_xcd < cu_slots ? per_xcd : cu_slots)));
    constexpr int NBUF = ring_depth<1, 1, BN, TH, TW, NK>();
    if (a.canvas_gx)
        hipLaunchKernelGGL((conv_bf16_ring_kernel<1, 1, BN, WM, WN, TH, TW, NK, NBUF, false, false, !CAT, DT>), pgrid, dim3(512), 0, st, a, (int)nblk);
    else
        hipLaunchKernelGGL((conv_bf16_ring_kernel<1, 1, BN, WM, WN, TH, TW, NK, NBUF, false, CAT, false, DT>), pgrid, dim3(512), 0, st, a, (int)nblk);
    AY_CHECK_LAUNCH("conv_bf16_ring_kernel<1x1>");
    return AY_OK;
}

// Data gradient of a 3x3 stride-2 convolution as four 2x2-window stride-1 convolutions over dz, one per parity class of the
// output pixel: the ring kernel with KS = 2, classes riding in the channel-group index (class fastest, so the four classes of a
// tile -- which interleave in the same 128-byte lines of dx -- run side by side)
template <int BN, int WM, int WN, int NK, int TH = 8, bool PAIR = false>
static int launch_dgrad_s2(const ay_conv_desc* d, const void* dz, const void* w, const float* scale, const float* shift,
                           const void* residual, void* dx, int cin_pad, hipStream_t st) {
    constexpr int TW = 32;
    ay_conv_desc dd = *d;
    dd.cin = d->cout_pad;     // reduction over dz's channel planes
    dd.cout = d->cin;
    dd.cout_pad = cin_pad;
    dd.hin = d->hout, dd.win = d->wout;
    dd.ksize = 2, dd.stride = 1, dd.leaky = 0, dd.out_f32 = 0;
    ConvArgs a;
    fill_args(a, &dd, dz, w, scale, shift, residual, dx, TH, TW, BN);
    // BN is the TILE width: with PAIR it holds the two column-parity classes of BN / 2 real channels, and an item's group index carries
    // only the row parity
    a.n_cgroups = PAIR ? 2 * (cin_pad / (BN / 2)) : 4 * (cin_pad / BN);
    a.w_class_stride = (unsigned)((size_t)(d->cout_pad / 16) * 4 * 2 * cin_pad * 16);
    const long long nblk = (long long)a.tiles_x * a.tiles_y * d->batch * a.n_cgroups;
    if (nblk <= 0 || nblk > 0x7fffffffLL) {
        set_error("dgrad grid out of range (%lld)", nblk);
        return AY_ERR_ARG;
    }
    a.deal = next_deal_set(st);
    const int per_xcd = (int)((nblk + 7) / 8);
    const int cu_slots = conv_num_cus() / 8;
    dim3 pgrid((unsigned)(8 * (per_xcd < cu_slots ? per_xcd : cu_slots))), block(512);
    constexpr int NBUF = ring_depth<2, 1, BN, TH, TW, NK>();
    if (residual)
        hipLaunchKernelGGL((conv_bf16_ring_kernel<2, 1, BN, WM, WN, TH, TW, NK, NBUF, true, false, false, Bf16, PAIR>), pgrid, block, 0, st, a, (int)nblk);
    else
        hipLaunchKernelGGL((conv_bf16_ring_kernel<2, 1, BN, WM, WN, TH, TW, NK, NBUF, false, false, false, Bf16, PAIR>), pgrid, block, 0, st, a, (int)nblk);
    AY_CHECK_LAUNCH("conv_bf16_ring_kernel(dgrad s2)");
    return AY_OK;
}

}  // namespace ay

namespace ay {
template <typename DT>
static int conv1x1_cat_fwd(const ay_conv_desc* d, const void* src1_halfres, int c1, const void* src2, const void* w_packed,
                           const float* scale, const float* shift, void* out, ay_stream_t stream) {
    AY_CHECK_ARG(d && src1_halfres && src2 && w_packed && scale && shift && out, "ay_conv1x1_cat_fwd_bf16: null argument");
    AY_CHECK_ARG(d->ksize == 1 && d->stride == 1 && !d->out_f32, "ay_conv1x1_cat_fwd_bf16: 1x1 stride-1 bf16 only");
    AY_CHECK_ARG(c1 > 0 && c1 % 64 == 0 && d->cin > c1 && (d->cin - c1) % 64 == 0, "ay_conv1x1_cat_fwd_bf16: channel split %d + %d",
                 c1, d->cin - c1);
    AY_CHECK_ARG(d->cout_pad % 128 == 0 && d->cout_pad >= d->cout, "ay_conv1x1_cat_fwd_bf16: cout_pad %d (multiple of 128)", d->cout_pad);
    AY_CHECK_ARG(d->hin % 2 == 0 && d->win % 2 == 0 && d->hout == d->hin && d->wout == d->win, "ay_conv1x1_cat_fwd_bf16: even sizes");
    return launch_ring1x1<128, 2, 4, true, DT>(d, src1_halfres, c1, src2, w_packed, scale, shift, out, S(stream));
}
}  // namespace ay

extern "C" int ay_conv1x1_cat_fwd_bf16(const ay_conv_desc* d, const void* src1_halfres, int c1, const void* src2, const void* w_packed,
                                       const float* scale, const float* shift, void* out, ay_stream_t stream) {
    return ay::conv1x1_cat_fwd<ay::Bf16>(d, src1_halfres, c1, src2, w_packed, scale, shift, out, stream);
}
extern "C" int ay_conv1x1_cat_fwd_f16(const ay_conv_desc* d, const void* src1_halfres, int c1, const void* src2, const void* w_packed,
                                      const float* scale, const float* shift, void* out, ay_stream_t stream) {
    return ay::conv1x1_cat_fwd<ay::F16>(d, src1_halfres, c1, src2, w_packed, scale, shift, out, stream);
}

extern "C" int ay_conv_dgrad_s2_bf16(const ay_conv_desc* d, const void* dz, const void* w_s2_packed, const float* ones, const float* zeros,
                                     const void* residual, void* dx, int cin_pad, ay_stream_t stream) {
    using namespace ay;
    AY_CHECK_ARG(d && dz && w_s2_packed && ones && zeros && dx, "ay_conv_dgrad_s2_bf16: null argument");
    AY_CHECK_ARG(d->ksize == 3 && d->stride == 2 && d->hin == 2 * d->hout && d->win == 2 * d->wout,
                 "ay_conv_dgrad_s2_bf16: a 3x3 stride-2 convolution of an even-sized input (%dx%d -> %dx%d)", d->hin, d->win, d->hout, d->wout);
    AY_CHECK_ARG(d->cout_pad % 32 == 0 && cin_pad % 32 == 0 && cin_pad >= d->cin, "ay_conv_dgrad_s2_bf16: channels %d(%d) <- %d", d->cin, cin_pad,
                 d->cout_pad);
    AY_CHECK_ARG((long long)d->hin * d->win * 2 * cin_pad < (1ll << 31), "ay_conv_dgrad_s2_bf16: one image of dx exceeds 2 GiB");
    hipStream_t st = S(stream);
    const int kin = d->cout_pad;
    // Class pairs (AY_S2_PAIR, default on): one workgroup computes BOTH column-parity classes of a row parity -- the tile's channels are
    // the two classes side by side -- so that a wave stores neighbouring pixels in consecutive instructions (whole lines reach HBM).
    static const int pair = getenv("AY_S2_PAIR") ? atoi(getenv("AY_S2_PAIR")) : 1;
    // (the layers with 128 and more channels are compute-bound: a 256-wide pair tile leaves LDS for 16-channel stages only and
    // measured 2-5 % slower than one class per workgroup; they keep the class form)
    if (pair && kin % 32 == 0 && cin_pad % 128 != 0) {
        static const int pth16 = getenv("AY_S2_PAIR_TH16") ? atoi(getenv("AY_S2_PAIR_TH16")) : 1;   // 16x32-pixel items for the 32-channel layer
        if (pth16 && d->hout >= 16 && cin_pad % 64 != 0) return launch_dgrad_s2<64, 1, 8, 2, 16, true>(d, dz, w_s2_packed, ones, zeros, residual, dx, cin_pad, st);
        if (cin_pad % 64 == 0) return launch_dgrad_s2<128, 2, 4, 2, 8, true>(d, dz, w_s2_packed, ones, zeros, residual, dx, cin_pad, st);
        return launch_dgrad_s2<64, 1, 8, 2, 8, true>(d, dz, w_s2_packed, ones, zeros, residual, dx, cin_pad, st);
    }
    if (cin_pad % 128 == 0 && kin % 32 == 0) return launch_dgrad_s2<128, 2, 4, 2>(d, dz, w_s2_packed, ones, zeros, residual, dx, cin_pad, st);
    // the narrow layers (32 / 64 channels of dx: the first two stride-2 layers) have one or two stages per item and are bound by the
    // per-item cost of the ring kernel: 16x32-pixel items (half as many): 72.5 -> 72.0 ms per training step at B=32 / 1024^2
    // (AY_S2_TH16=0: 8x32 items)
    static const int th16 = getenv("AY_S2_TH16") ? atoi(getenv("AY_S2_TH16")) : 1;
    if (th16 && d->hout >= 16) {
        if (cin_pad % 128 != 0 && cin_pad % 64 == 0 && kin % 32 == 0) return launch_dgrad_s2<64, 1, 8, 2, 16>(d, dz, w_s2_packed, ones, zeros, residual, dx, cin_pad, st);
        if (cin_pad % 64 != 0 && kin % 32 == 0) return launch_dgrad_s2<32, 1, 8, 2, 16>(d, dz, w_s2_packed, ones, zeros, residual, dx, cin_pad, st);
    }
    if (cin_pad % 64 == 0 && kin % 32 == 0) return launch_dgrad_s2<64, 1, 8, 2>(d, dz, w_s2_packed, ones, zeros, residual, dx, cin_pad, st);
    if (kin % 64 == 0) return launch_dgrad_s2<32, 1, 8, 4>(d, dz, w_s2_packed, ones, zeros, residual, dx, cin_pad, st);
    return launch_dgrad_s2<32, 1, 8, 2>(d, dz, w_s2_packed, ones, zeros, residual, dx, cin_pad, st);
}

namespace ay {
template <typename DT>
static int conv_fwd_16(const ay_conv_desc* d, const void* src, const void* w_packed, const float* scale,
                       const float* shift, const void* residual, void* out, ay_stream_t stream) {
    AY_CHECK_ARG(d && src && w_packed && scale && shift && out, "ay_conv_fwd_bf16: null argument");
    AY_CHECK_ARG(d->ksize == 1 || d->ksize == 3, "ay_conv_fwd_bf16: ksize %d unsupported", d->ksize);
    AY_CHECK_ARG(d->stride == 1 || (d->stride == 2 && d->ksize == 3), "ay_conv_fwd_bf16: stride %d unsupported", d->stride);
    AY_CHECK_ARG(d->cin % 16 == 0 && d->cout_pad % 32 == 0 && d->cout_pad >= d->cout, "ay_conv_fwd_bf16: channels %d->%d(%d)",
                 d->cin, d->cout, d->cout_pad);
    const int pad = (d->ksize - 1) / 2;
    AY_CHECK_ARG(d->hout == (d->hin + 2 * pad - d->ksize) / d->stride + 1 && d->wout == (d->win + 2 * pad - d->ksize) / d->stride + 1,
                 "ay_conv_fwd_bf16: output size mismatch");
    AY_CHECK_ARG(!(d->out_f32 && residual), "ay_conv_fwd_bf16: f32 output has no residual form");
    AY_CHECK_ARG((long long)d->hout * d->wout * 2 * d->cout_pad < (1ll << 31),
                 "ay_conv_fwd_bf16: one image's output (%dx%dx%d) exceeds the 2 GiB a store descriptor addresses", d->hout, d->wout,
                 d->cout_pad);
    hipStream_t st = S(stream);
    const int cp = d->cout_pad;
    if (d->ksize == 3 && d->stride == 1) {
        AY_CHECK_ARG(!d->out_f32, "ay_conv_fwd_bf16: 3x3 f32 output unsupported");
        static const int tile16 = getenv("AY_TILE16") ? atoi(getenv("AY_TILE16")) : 1;
        static const int m16 = getenv("AY_M16") ? atoi(getenv("AY_M16")) : 1;
        int ctx = 0, cty = 0;
        if (cp % 128 == 0 && tile16 && m16 && d->hout >= 16 && d->cin % 32 == 0 && canvas_plan(d, 16, 32, &ctx, &cty) == 0)
            return DT::id == AY_DT_F16 ? ay_conv3x3_m16_fwd_f16(d, src, w_packed, scale, shift, residual, out, stream)
                                       : ay_conv3x3_m16_fwd_bf16(d, src, w_packed, scale, shift, residual, out, stream);  // v_mfma_f32_16x16x32_{bf16,f16}
        if (cp % 128 == 0 && tile16 && d->hout >= 16)
            return launch<3, 1, 128, 2, 4, 16, 32, 1, false, true, DT>(d, src, w_packed, scale, shift, residual, out, st);
        if (cp % 128 == 0) return launch<3, 1, 128, 2, 4, 8, 32, 1, false, true, DT>(d, src, w_packed, scale, shift, residual, out, st);
        if (cp % 64 == 0) return launch<3, 1, 64, 1, 8, 8, 32, 1, false, true, DT>(d, src, w_packed, scale, shift, residual, out, st);
        // 32 output channels (the stem and the data gradients that end in it during training: HBM-bound, 1.4 ms per launch at
        // B=32 / 1024^2 on the register-staged kernel): the ring kernel with a 32-channel tile, one 32x32 block per wave
        return launch<3, 1, 32, 1, 8, 8, 32, 1, false, true, DT>(d, src, w_packed, scale, shift, residual, out, st);
    }
    if (d->ksize == 3 && d->stride == 2) {
        AY_CHECK_ARG(!d->out_f32, "ay_conv_fwd_bf16: 3x3 f32 output unsupported");
        if (cp % 128 == 0) return launch<3, 2, 128, 2, 4, 8, 32, 1, false, true, DT>(d, src, w_packed, scale, shift, residual, out, st);
        if (cp % 64 == 0) return launch<3, 2, 64, 1, 8, 8, 32, 1, false, true, DT>(d, src, w_packed, scale, shift, residual, out, st);
        return launch<3, 2, 32, 1, 4, 4, 32, 1, false, false, DT>(d, src, w_packed, scale, shift, residual, out, st);
    }
    // 1x1
    if (d->cin % 64 == 0) {
        if (d->out_f32) {
            if (cp % 64 == 0) return launch<1, 1, 64, 1, 4, 8, 32, 4, true, false, DT>(d, src, w_packed, scale, shift, residual, out, st);
            return launch<1, 1, 32, 1, 4, 8, 32, 4, true, false, DT>(d, src, w_packed, scale, shift, residual, out, st);
        }
        static const int bn256 = getenv("AY_BN256") ? atoi(getenv("AY_BN256")) : 1;
        if (cp % 256 == 0 && !residual && bn256)
            return launch_ring1x1<256, 4, 2, false, DT>(d, nullptr, 0, src, w_packed, scale, shift, out, st);
        if (cp % 128 == 0) return launch<1, 1, 128, 2, 4, 8, 32, 4, false, true, DT>(d, src, w_packed, scale, shift, residual, out, st);
        if (cp % 64 == 0) return launch<1, 1, 64, 1, 8, 8, 32, 4, false, true, DT>(d, src, w_packed, scale, shift, residual, out, st);
        return launch<1, 1, 32, 1, 8, 8, 32, 4, false, true, DT>(d, src, w_packed, scale, shift, residual, out, st);
    }
    if (d->out_f32) return launch<1, 1, 32, 1, 4, 8, 32, 1, true, false, DT>(d, src, w_packed, scale, shift, residual, out, st);
    return launch<1, 1, 32, 1, 4, 8, 32, 1, false, false, DT>(d, src, w_packed, scale, shift, residual, out, st);
}
}  // namespace ay

namespace ay {
// A detection head with its decode: the linear 1x1 convolution of models.py:33-40 (heads 81 / 93 / 105: bias, no BatchNorm, no
// activation) through conv_bf16_kernel<..., DECODE>, whose epilogue writes rows [row_offset, row_offset + A G G) of pred.
template <typename DT>
static int head_decode_fwd(const ay_conv_desc* d, const void* src, const void* w_packed, const float* scale, const float* shift,
                           int num_anchors, int num_classes, int img_dim, const float* anchors_wh, float* pred, int n_total,
                           int row_offset, ay_stream_t stream) {
    AY_CHECK_ARG(d && src && w_packed && scale && shift && anchors_wh && pred, "ay_head_decode_fwd: null argument");
    AY_CHECK_ARG(d->ksize == 1 && d->stride == 1 && !d->leaky && d->cin % 16 == 0, "ay_head_decode_fwd: a linear 1x1 block (cin %% 16 == 0)");
    AY_CHECK_ARG(d->hout == d->hin && d->wout == d->win && d->hout == d->wout, "ay_head_decode_fwd: square grid, same size in and out");
    const int K = 5 + num_classes;
    AY_CHECK_ARG(num_anchors >= 1 && num_anchors <= 6 && num_classes >= 1 && num_anchors * K == d->cout && d->cout_pad == (d->cout + 31) / 32 * 32,
                 "ay_head_decode_fwd: %d anchors x (5 + %d classes) != %d filters (padded %d)", num_anchors, num_classes, d->cout, d->cout_pad);
    AY_CHECK_ARG(row_offset >= 0 && row_offset + num_anchors * d->hout * d->wout <= n_total, "ay_head_decode_fwd: rows out of range");
    ConvArgs a;
    fill_args(a, d, src, w_packed, scale, shift, nullptr, nullptr, 8, 32, 32);
    DecodeArgs dd{};
    dd.pred = pred;
    dd.n_total = n_total;
    dd.row_offset = row_offset;
    dd.A = num_anchors;
    dd.K = K;
    dd.stride = (float)((double)img_dim / (double)d->hout);   // Python float division, then cast (models.py:119): as ay_yolo_decode
    for (int i = 0; i < num_anchors; ++i) dd.aw[i] = anchors_wh[2 * i], dd.ah[i] = anchors_wh[2 * i + 1];
    const long long nblk = (long long)a.tiles_x * a.tiles_y * d->batch * a.n_cgroups;
    AY_CHECK_ARG(nblk > 0 && nblk <= 0x7fffffffLL, "ay_head_decode_fwd: grid out of range");
    dim3 grid((unsigned)nblk), block(256);
    if (d->cin % 64 == 0)
        hipLaunchKernelGGL((conv_bf16_kernel<1, 1, 32, 1, 4, 8, 32, 4, true, false, DT, true>), grid, block, 0, S(stream), a, dd);
    else
        hipLaunchKernelGGL((conv_bf16_kernel<1, 1, 32, 1, 4, 8, 32, 1, true, false, DT, true>), grid, block, 0, S(stream), a, dd);
    AY_CHECK_LAUNCH("conv_bf16_kernel<DECODE>");
    return AY_OK;
}
}  // namespace ay

extern "C" int ay_head_decode_fwd_bf16(const ay_conv_desc* d, const void* src, const void* w_packed, const float* scale, const float* shift,
                                       int num_anchors, int num_classes, int img_dim, const float* anchors_wh, float* pred, int n_total,
                                       int row_offset, ay_stream_t stream) {
    return ay::head_decode_fwd<ay::Bf16>(d, src, w_packed, scale, shift, num_anchors, num_classes, img_dim, anchors_wh, pred, n_total, row_offset, stream);
}
extern "C" int ay_head_decode_fwd_f16(const ay_conv_desc* d, const void* src, const void* w_packed, const float* scale, const float* shift,
                                      int num_anchors, int num_classes, int img_dim, const float* anchors_wh, float* pred, int n_total,
                                      int row_offset, ay_stream_t stream) {
    return ay::head_decode_fwd<ay::F16>(d, src, w_packed, scale, shift, num_anchors, num_classes, img_dim, anchors_wh, pred, n_total, row_offset, stream);
}

extern "C" int ay_conv_fwd_bf16(const ay_conv_desc* d, const void* src, const void* w_packed, const float* scale,
                                const float* shift, const void* residual, void* out, ay_stream_t stream) {
    return ay::conv_fwd_16<ay::Bf16>(d, src, w_packed, scale, shift, residual, out, stream);
}
extern "C" int ay_conv_fwd_f16(const ay_conv_desc* d, const void* src, const void* w_packed, const float* scale,
                               const float* shift, const void* residual, void* out, ay_stream_t stream) {
    return ay::conv_fwd_16<ay::F16>(d, src, w_packed, scale, shift, residual, out, stream);
}
