// 3x3 stride-1 convolution, 128 output channels x 16x32 pixels per workgroup, on v_mfma_f32_16x16x32_bf16.
//
// Same persistent all-DMA ring structure as conv_bf16_ring_kernel<3,1,128,2,4,16,32,1,2> (ay_conv_bf16.hip: LDS-DMA staging with
// counted waits, items dealt per XCD through atomic counters and an LDS mailbox, fused BN-affine + LeakyReLU (+ shortcut)
// epilogue; reference models.py:26-45, 246-248) with the matrix instruction changed: on non-trivial operands the chip holds
// a higher clock on the 16x16x32 shape than on 32x32x16 (scripts/micro/mfma_shape.hip on this kernel's wave tile, every
// operand re-read from LDS: 1 773 vs 1 610 TFLOP/s on network-like data, 1 814 vs 1 637 on uniform random, 2 243 vs 2 392 on
// zeros -- MI355X_MICROARCH.md "DVFS give-back" item 7).
//
// K = 32 of one MFMA = two filter taps x the 16 input channels of a stage (activations are [C/16] planes, a stage buffer of 32
// channels would not fit twice in LDS beside the 128-channel filter slab):
//     lane (r = lane & 15, g = lane >> 4):  k-group g = (tap of the pair: g >> 1, channel half: g & 1)
// With the taps paired (0,1) (2,3) (4,5) (6,7) the four k-groups of the FILTER operand are four consecutive [tap][half] slabs of
// the stage's LDS image (the packed filter format is unchanged); the PIXEL operand's second tap is a per-lane shift of 1 pixel
// ((0,1) (4,5) (6,7)) or 32 pixels ((2,3): tap (0,2) -> (1,0)).  Tap 8 is left over in every stage; two consecutive stages
// share one MFMA for it ("straddle": lanes g < 2 read tap 8 of the even stage's ring slot, lanes g >= 2 tap 8 of the odd
// stage's), issued when both slots are resident, i.e. right after the odd stage has landed; one more barrier then frees the
// even slot for the DMA of the next stage.  Per two stages: 4 + 1 + 4 K32-steps of 32 MFMAs per wave -- the same MFMA cycles
// as 18 taps x 8 MFMAs of 32x32x16 -- and 3 barriers instead of 2.  The input channel count must be a multiple of 32.
//
// Wave tile as before: 64 channels x 128 pixels (4 x 8 accumulator tiles of 16x16), 8 waves = 2 (channels) x 4 (pixel rows).
// Fragments: the filter fragments of the next step are prefetched into a second register set; the pixel fragments rotate
// through four registers sets (tile n+4 is loaded right behind the 4 MFMAs of tile n), so 48 fragment registers serve the 32
// MFMAs of a step -- with all 8 pixel fragments resident the kernel spilled (128 accumulators + 64 + addressing > 256).
// LDS pixel image [half][624 px][16 B]: 624 = 18 x 34 rounded up to a multiple of 16 keeps the two halves' lanes of one
// ds_read_b128 lane group on disjoint banks.
// C/D layout of 16x16: col (pixel) = lane & 15, row (channel) = 4 * (lane >> 4) + reg.
#include <stdlib.h>

#include <type_traits>

#include "ay_conv_common.h"

namespace ay {

#ifndef AY_M16_PRIO
#define AY_M16_PRIO 1
#endif
#ifdef AY_PHASE_CLOCK
// instrumented build only (AY_PHASE_CLOCK=1 python build.py, AY_DBG=8 at run time): 100 MHz ticks of wave 0, summed over
// workgroups: [0] stage loops, [1] epilogues, [2] items, [3] workgroups, [4] whole kernel per workgroup, [5] waits for landed
// stages (vmcnt + barrier, both kinds), [6] slowest workgroup, [7] straddle step + its barrier
__device__ unsigned long long g_phase_ticks_m16[8];
#define AY_CLK(...) __VA_ARGS__
#else
#define AY_CLK(...)
#endif

template <bool HAS_RES, typename DT = Bf16>
__global__ void __launch_bounds__(512, 2) conv3x3_m16_ring_kernel(ConvArgs a, int n_items) {
    DT::enter();
    typedef typename DT::vec8 vec8;
    constexpr int BN = 128, TH = 16, TW = 32;
    constexpr int IN_W = TW + 2, IN_PIX = (TH + 2) * IN_W;  // 18 x 34 halo tile
    constexpr int IN_PIXP = (IN_PIX + 15) / 16 * 16;          // 624: half-plane stride, a multiple of 16 pixels (banks)
    constexpr int PX_PIECES = (2 * IN_PIXP + 63) / 64;        // 20 one-KiB DMA pieces per 16-channel pixel slab
    constexpr int PIX_SLAB = PX_PIECES * 1024;
    constexpr int W_PIECES = 9 * 2 * BN * 16 / 1024;          // 36
    constexpr int W_SLAB = W_PIECES * 1024;
    constexpr int W_BASE = PIX_SLAB;
    constexpr int BUF_BYTES = PIX_SLAB + W_SLAB;              // 56 KiB per ring slot, two slots
    constexpr int PWP = (PX_PIECES + 7) / 8;                  // 3 pixel pieces per wave and stage
    constexpr int PWW = (W_PIECES + 2 + 7) / 8;               // 5 filter-side pieces (36 filter pieces + scale piece + shift piece)
    constexpr int PW = PWP + PWW;                             // 8
    constexpr int DUMMY_BASE = 2 * BUF_BYTES;
    constexpr int SS_BASE = DUMMY_BASE + 1024;                // 4 x 2 KiB [scale 128 | 0][shift 128 | 0], by item sequence & 3
    constexpr int MBOX_BASE = SS_BASE + 4 * 2048;
    constexpr int LDS_BYTES = MBOX_BASE + 64;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS ring");
    constexpr int MT = 4, NT = 8;                             // 16x16 accumulator tiles per wave: 64 channels x 128 pixels

    __shared__ __attribute__((aligned(16))) uint8_t lds[LDS_BYTES];
    __builtin_amdgcn_s_setprio(2);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int r = lane & 15, g = lane >> 4, h = g & 1, hi = g >> 1;

    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int per_xcd = (n_items + 7) >> 3;
    const int first = xcd * per_xcd;
    const int last = min(first + per_xcd, n_items);
    int item = first + slot;
    // ---- item dealing: as conv_bf16_ring_kernel (static first items, then the XCD's atomic counter through an LDS mailbox)
    const bool dyn = a.deal != nullptr;
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
    typedef volatile __attribute__((address_space(3))) int lds_vint;
    lds_vint* mbox = (lds_vint*)(__attribute__((address_space(3))) int*)(lds + MBOX_BASE);
    constexpr int D = 3;  // fetch-ahead distance in items
    auto fetch_id = [&](int prev) __attribute__((always_inline)) -> int {  // thread 0 only
        if (prev >= last) return last;
        if (!dyn) return prev + slots;
        // whatever the counter holds, the id stays inside this XCD's range or reads as "no more items"
        const unsigned n = atomicAdd(a.deal + xcd, 1u);
        return n < (unsigned)(last - first) ? first + D * slots + (int)n : last;
    };
    if (tid < D) mbox[tid] = min(item + tid * slots, last);
    __syncthreads();
    int seq_l = 0, seq_c = 0;

    const size_t in_plane = (size_t)a.hin * a.win * 32;
    const int CP = a.cout_pad;
    const size_t w_stage_stride = (size_t)9 * 2 * CP * 16;
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const unsigned lds_base = lds_addr_of(lds);
    const int nstages = a.cin / 16;  // even (host-checked)

    // ---- loader (piece i of this wave = global piece i*8 + wave of its kind) -------------------------------------
    // Every piece is a `buffer_load_dwordx4 ... lds`: a wave-uniform descriptor + scalar offset (stage, tap) and ONE 32-bit
    // per-lane offset; a lane that has nothing to fetch (halo pixel outside the image, padding of the piece lists) carries the
    // offset 0x80000000, which is out of range for every descriptor and reads as zero.  No 64-bit per-lane addresses, no zero
    // page, no selects: the 3 pixel offsets of a wave are all the per-lane state the loader keeps across an item.
    constexpr unsigned OOB = 0x80000000u;
    unsigned px_off[PWP];
    const uint8_t* ld_src = nullptr;  // image b of the input
    int ld_cg = 0;
    int ld_item = item, ld_s = 0, ld_par = 0;
    bool ld_done = false;
    auto setup_loader = [&](int it) __attribute__((always_inline)) {
        ld_cg = it % a.n_cgroups;
        const int pt = it / a.n_cgroups;
        const int b = pt / tiles_per_img;
        const int y0 = ((pt / a.tiles_x) % a.tiles_y) * TH, x0 = (pt % a.tiles_x) * TW;
        ld_src = a.src + (size_t)b * (a.cin / 16) * in_plane;
#pragma unroll
        for (int i = 0; i < PWP; ++i) {
            const int q = i * 8 + wave;
            const int u = q * 64 + lane;  // unit inside the slab, LDS order [half][IN_PIXP]
            const int hh = u / IN_PIXP, P = u % IN_PIXP;
            const int iy = y0 - 1 + P / IN_W, ix = x0 - 1 + P % IN_W;
            const bool in = q < PX_PIECES && hh < 2 && P < IN_PIX && iy >= 0 && iy < a.hin && ix >= 0 && ix < a.win;
            px_off[i] = in ? (unsigned)((iy * a.win + ix) * 32 + hh * 16) : OOB;
        }
    };
    auto issue_piece = [&](int i, int buf) __attribute__((always_inline)) {
        if (i < PWP) {
            const int q = i * 8 + wave;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(ld_src), 0, (int)((a.cin / 16) * in_plane), 0x00020000);
            const int dst = q < PX_PIECES ? buf * BUF_BYTES + q * 1024 : DUMMY_BASE;
            dma16_buf(rs, px_off[i], (unsigned)(ld_s * (int)in_plane), lds_base + dst);
        } else {
            const int qq = (i - PWP) * 8 + wave;  // wave-uniform: filter piece | scale piece | shift piece | padding
            if (qq < W_PIECES) {
                // filter image of the stage [tap][half][BN][16 B] = 36 pieces of 64 rows: piece qq = slab qq/2, rows (qq&1)*64..
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.w), 0, 0x7fffffff, 0x00020000);
                const unsigned so = (unsigned)(ld_s * (int)w_stage_stride + ((qq >> 1) * CP + ld_cg * BN + (qq & 1) * 64) * 16);
                dma16_buf(rs, (unsigned)lane * 16u, so, lds_base + buf * BUF_BYTES + W_BASE + qq * 1024);
            } else {
                // lanes 0..31 carry 4 scales (piece W_PIECES) or 4 shifts (piece W_PIECES + 1) each; region [scale 512 B | 0][shift 512 B | 0]
                const bool ss = qq < W_PIECES + 2;
                const float* base = qq == W_PIECES ? a.scale : a.shift;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, CP * 4, 0x00020000);
                const unsigned vo = (ss && lane < 32) ? (unsigned)(ld_cg * BN * 4 + lane * 16) : OOB;
                dma16_buf(rs, vo, 0u, lds_base + (ss ? SS_BASE + ld_par * 2048 + (qq - W_PIECES) * 1024 : DUMMY_BASE));
            }
        }
    };
    auto advance_loader = [&]() __attribute__((always_inline)) {
        if (++ld_s == nstages) {
            ld_s = 0;
            ld_par = (ld_par + 1) & 3;
            ++seq_l;
            ld_item = __builtin_amdgcn_readfirstlane(mbox[seq_l & 7]);
            if (ld_item < last)
                setup_loader(ld_item);
            else
                ld_done = true;
        }
    };

    // ---- fragment addresses (per lane; tile positions and taps are immediates) ---------------------------------------
    const int pb = (h * IN_PIXP + wn * 4 * IN_W + r) * 16;             // pixel (tile row 4 wn, column r), this lane's half
    const int pb1 = pb + hi * 16;                                      // second tap of the pair one pixel to the right
    const int pb32 = pb + hi * 32 * 16;                                // pair (2,3): tap (0,2) -> tap (1,0) = +IN_W - 2 pixels
    const int pbs = pb + hi * BUF_BYTES;                               // straddle: second half of K from the odd slot
    const int wa = W_BASE + (g * BN + wm * 64 + r) * 16;               // four consecutive [tap][half] slabs = the four k-groups
    const int was = W_BASE + ((16 + h) * BN + wm * 64 + r) * 16 + hi * BUF_BYTES;  // tap 8 of the even | odd slot

    // ---- prologue: stage 0 landed in slot 0 ------------------------------------------------------------------------------
    setup_loader(item);
#pragma unroll
    for (int i = 0; i < PW; ++i) issue_piece(i, 0);
    advance_loader();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    int par = 0;
    AY_CLK(const bool clk = (a.dbg & 8) && wave == 0; unsigned long long tk_stage = 0, tk_epi = 0, tk_items = 0, tk_wait = 0, tk_str = 0, tk0 = 0, tkw = 0;
           const unsigned long long tk_begin = wall_clock64();)
    while (true) {
        AY_CLK(if (clk) tk0 = wall_clock64();)
        const int cg = item % a.n_cgroups;
        const int pt = item / a.n_cgroups;
        const int b = pt / tiles_per_img;
        const int y0 = ((pt / a.tiles_x) % a.tiles_y) * TH, x0 = (pt % a.tiles_x) * TW;
        const int next_item = __builtin_amdgcn_readfirstlane(mbox[(seq_c + 1) & 7]);
        const bool has_next = next_item < last;

        f32x4 acc[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

        // ---- epilogue state of this item (defined here: part of the residual is requested from inside the last stage pair)
        typedef __attribute__((address_space(3))) const f32x4 lds_f4;
        lds_f4* sl4 = (lds_f4*)(__attribute__((address_space(3))) const float*)(lds + SS_BASE + par * 2048);
        const float slope = a.leaky ? 0.1f : 1.0f;
        const size_t out_plane_px = (size_t)a.hout * a.wout;
        const unsigned plane_bytes = (unsigned)out_plane_px * 32u;
        const unsigned img_bytes = plane_bytes * (unsigned)(CP / 16);
        const int b0 = __builtin_amdgcn_readfirstlane(b);
        const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)b0 * img_bytes, 0, (int)img_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint8_t*>(HAS_RES ? a.residual : a.out) + (size_t)b0 * img_bytes, 0, (int)img_bytes, 0x00020000);
        const int cbase = cg * BN + wm * 64;  // first channel of this wave
        const int col = (g & 1) * 16 + r;     // pixel column inside the tile row, byte half (g >> 1) * 16 of the 32-byte pixel
        constexpr int NU = MT * (NT / 2);     // 16 store units
#ifndef AY_M16_RD
#define AY_M16_RD 8
#endif
#ifndef AY_M16_EARLY
#define AY_M16_EARLY 0
#endif
        // residual units requested inside the last stage pair: measured SLOWER (4 units: 9 VGPRs spilled in the stage loop, 0.479 ->
        // 0.485 ms per launch; 8 units: 28 spilled, 0.516 ms) -- the loop sits at ~246 of 256 registers -- so 0; look-ahead inside the
        // epilogue 4 / 8 / 16 units: 0.468 / 0.468 / 0.472 ms (same-box A/B): the residual epilogue is not a latency chain, it is the
        // HBM burst of all CUs reading residual and writing output at the same time
        constexpr int EARLY = HAS_RES ? AY_M16_EARLY : 0;
        constexpr int RD = AY_M16_RD;         // residual look-ahead in units inside the epilogue
        u32x4 rres[HAS_RES ? NU : 1];
        u32x4 outv[HAS_RES ? NU : 1];
        auto unit_off = [&](int np, bool clamp, bool& ok) __attribute__((always_inline)) -> unsigned {
            int oy = y0 + wn * 4 + np, ox = x0 + col;
            ok = oy < a.hout && ox < a.wout;
            if (clamp) oy = min(oy, a.hout - 1), ox = min(ox, a.wout - 1);
            return ((unsigned)oy * a.wout + ox) * 32u + (unsigned)(g >> 1) * 16u;
        };
        auto load_res = [&](int t) __attribute__((always_inline)) {
            const int np = t / MT, m = t % MT;
            bool ok;
            const unsigned vo = unit_off(np, true, ok);  // clamped: always a valid address, loads stay unconditional
            const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)((cbase + m * 16) >> 4) * plane_bytes);
            rres[t] = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, vo, so, 0);
        };

        vec8 fa[2][MT], fb[4];
        // kind 0..3: pair j of the slot `sl`; kind 4: the straddle step (both slots)
        auto ld_a = [&](int kind, int sl, int m) __attribute__((always_inline)) -> vec8 {
            if (kind == 4) return *reinterpret_cast<const vec8*>(lds + was + m * 256);
            return *reinterpret_cast<const vec8*>(lds + sl * BUF_BYTES + wa + kind * (4 * BN * 16) + m * 256);
        };
        auto ld_b = [&](int kind, int sl, int n) __attribute__((always_inline)) -> vec8 {
            const int tile = ((n >> 1) * IN_W + (n & 1) * 16) * 16;
            if (kind == 4) return *reinterpret_cast<const vec8*>(lds + pbs + tile + (2 * IN_W + 2) * 16);
            const int ta = 2 * kind;
            const int tap = ((ta / 3) * IN_W + ta % 3) * 16;
            return *reinterpret_cast<const vec8*>(lds + sl * BUF_BYTES + (kind == 1 ? pb32 : pb1) + tile + tap);
        };
        // one K32-step: 32 MFMAs; behind the MFMAs of pixel tile n its fragment register takes the next step's tile n, the
        // filter fragments of the next step go to the other register set; PREFETCH = false where the next step's slot has not
        // landed yet (the step in front of a stage barrier): its fragments are then loaded after the barrier
        // DMA pieces ride behind the MFMA groups 1, 3, 5, 7 of a step: PLIST 0 = none, 1 = pieces 0..3, 2 = pieces 4..7,
        // 3 = the early pieces 3..6 (filter taps 0..7), 4 = the late pieces 0, 1, 2, 7 (pixels, tap 8, scale/shift)
        auto step = [&](auto KIND, auto SL, auto NKIND, auto NSL, auto CUR, auto PREFETCH, auto PLIST, bool dma, int dma_slot) __attribute__((always_inline)) {
            constexpr int kind = decltype(KIND)::value, sl = decltype(SL)::value, nkind = decltype(NKIND)::value, nsl = decltype(NSL)::value;
            constexpr int cur = decltype(CUR)::value;
            constexpr bool prefetch = decltype(PREFETCH)::value;
            constexpr int plist = decltype(PLIST)::value;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
#if AY_M16_PRIO
                __builtin_amdgcn_s_setprio(3);
#endif
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m][n] = DT::mfma16(fa[cur][m], fb[n & 3], acc[m][n]);
#if AY_M16_PRIO
                __builtin_amdgcn_s_setprio(2);
#endif
                if (n < 4) {
                    fb[n & 3] = ld_b(kind, sl, n + 4);
                    // the next step's first MFMA group needs all four filter fragments: they are requested in the first half of
                    // this step (behind groups 4..7 the last of them was waited for at every step boundary)
                    if constexpr (prefetch) fa[cur ^ 1][n] = ld_a(nkind, nsl, n);
                } else if constexpr (prefetch) {
                    fb[n & 3] = ld_b(nkind, nsl, n - 4);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (plist != 0) {
                    if (dma && (n & 1)) {
                        const int j = n >> 1;  // 0..3 (n is a constant after unrolling)
                        const int piece = plist == 1 ? j : plist == 2 ? 4 + j : plist == 3 ? 3 + j : (j < 3 ? j : 7);
                        issue_piece(piece, dma_slot);
                    }
                }
            }
        };
        auto load_all = [&](auto KIND, auto SL, auto CUR) __attribute__((always_inline)) {
            constexpr int kind = decltype(KIND)::value, sl = decltype(SL)::value, cur = decltype(CUR)::value;
#pragma unroll
            for (int m = 0; m < MT; ++m) fa[cur][m] = ld_a(kind, sl, m);
#pragma unroll
            for (int n = 0; n < 4; ++n) fb[n] = ld_b(kind, sl, n);
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        using I3 = std::integral_constant<int, 3>;
        using I4 = std::integral_constant<int, 4>;
        using T = std::true_type;
        using F = std::false_type;

        load_all(I0{}, I0{}, I0{});
        for (int s = 0; s < nstages; s += 2) {
            const bool last_pair = s + 2 == nstages;
            // ---- even stage (slot 0).  The odd stage of this item streams into slot 1, all of it issued in the first two steps:
            // an LDS-DMA piece lands ~1.1 us after its issue under load, about two steps; spread evenly over the stage the last
            // pieces were waited for 0.85 us at every stage end (phase clock: 27 % of the stage loop).
            step(I0{}, I0{}, I1{}, I0{}, I0{}, T{}, I1{}, true, 1);
            step(I1{}, I0{}, I2{}, I0{}, I1{}, T{}, I2{}, true, 1);
            step(I2{}, I0{}, I3{}, I0{}, I0{}, T{}, I0{}, false, 1);
            step(I3{}, I0{}, I4{}, I0{}, I1{}, F{}, I0{}, false, 1);
            advance_loader();
            AY_CLK(if (clk) tkw = wall_clock64();)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the odd stage has landed
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            AY_CLK(if (clk) { const unsigned long long t = wall_clock64(); tk_wait += t - tkw; tkw = t; })
            // ---- tap 8 of both stages in one K32-step.  Every wave is through with the filter taps 0..7 of slot 0 (barrier
            // above), the straddle reads tap 8 and the pixels only: the next stage's taps 0..7 stream into slot 0 already
            const bool issued = !ld_done;
            load_all(I4{}, I0{}, I0{});
            step(I4{}, I0{}, I0{}, I1{}, I0{}, T{}, I3{}, issued, 0);
            __builtin_amdgcn_s_barrier();  // every wave is through with slot 0: its pixels and tap 8 may follow
            asm volatile("" ::: "memory");
            AY_CLK(if (clk) tk_str += wall_clock64() - tkw;)
            // ---- odd stage (slot 1); the rest of stage s+2 of this item, or of stage 0 of the next, streams into slot 0
            step(I0{}, I1{}, I1{}, I1{}, I1{}, T{}, I4{}, issued, 0);
            if constexpr (HAS_RES && EARLY > 0) {
                // the first EARLY residual units are requested three steps (~1.4 us) before the epilogue: their round trip -- ~2 us
                // when every CU reaches its epilogue together -- otherwise opens the epilogue with nothing to do
                if (last_pair) {
#pragma unroll
                    for (int t = 0; t < EARLY; ++t) load_res(t);
                }
            }
            step(I1{}, I1{}, I2{}, I1{}, I0{}, T{}, I0{}, false, 0);
            step(I2{}, I1{}, I3{}, I1{}, I1{}, T{}, I0{}, false, 0);
            step(I3{}, I1{}, I0{}, I0{}, I0{}, F{}, I0{}, false, 0);
            if (issued) advance_loader();
            if (!(last_pair && !has_next)) {
                AY_CLK(if (clk) tkw = wall_clock64();)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next even stage has landed
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                AY_CLK(if (clk) tk_wait += wall_clock64() - tkw;)
                if (!last_pair) load_all(I0{}, I0{}, I0{});
            }
        }
        AY_CLK(if (clk) { const unsigned long long t = wall_clock64(); tk_stage += t - tk0; tk0 = t; })

        int fetched = last;
        if (tid == 0) fetched = fetch_id(mbox[(seq_c + D - 1) & 7]);

        // ---- epilogue: affine + leaky (+ residual) -> bf16 -> 16-byte stores --------------------------------------------
        // A store unit = (m, row pair np): the accumulator tiles (m, 2np) and (m, 2np+1) are the two halves of one 32-pixel
        // tile row; lane (r, q) holds channels 4q..4q+3 of pixel r of each.  One v_permlane16_swap per dword (rows of 16
        // lanes: odd rows of the first operand <-> even rows of the second) leaves lane (r, q) with channels 8(q>>1)..+7 of
        // pixel (q&1)*16 + r: a 16-byte store, 1 KiB contiguous per wave instruction.  The residual is loaded in that store
        // layout and brought to the accumulator layout by the same swaps (their own inverse).
        {
            if constexpr (HAS_RES) {
#pragma unroll
                for (int t = EARLY; t < RD; ++t) load_res(t);
            }
#pragma unroll
            for (int np = 0; np < NT / 2; ++np) {
                bool ok;
                const unsigned vo = unit_off(np, false, ok);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int t = np * MT + m;
                    const int l0 = wm * 64 + m * 16 + 4 * g;  // multiple of 4 floats
                    const f32x4 sc = sl4[l0 >> 2], sh = sl4[(256 + l0) >> 2];
                    const f32x4 X = acc[m][2 * np], Y = acc[m][2 * np + 1];
                    f32x2 x01 = f32x2{X[0], X[1]} * f32x2{sc[0], sc[1]} + f32x2{sh[0], sh[1]};
                    f32x2 x23 = f32x2{X[2], X[3]} * f32x2{sc[2], sc[3]} + f32x2{sh[2], sh[3]};
                    f32x2 y01 = f32x2{Y[0], Y[1]} * f32x2{sc[0], sc[1]} + f32x2{sh[0], sh[1]};
                    f32x2 y23 = f32x2{Y[2], Y[3]} * f32x2{sc[2], sc[3]} + f32x2{sh[2], sh[3]};
                    x01 = leaky2(x01, slope), x23 = leaky2(x23, slope), y01 = leaky2(y01, slope), y23 = leaky2(y23, slope);
                    if constexpr (HAS_RES) {
                        const u32x4 rv = rres[t];
                        auto s0 = __builtin_amdgcn_permlane16_swap(rv[0], rv[2], false, false);
                        auto s1 = __builtin_amdgcn_permlane16_swap(rv[1], rv[3], false, false);
                        x01 += DT::unpack2(s0[0]), x23 += DT::unpack2(s1[0]);
                        y01 += DT::unpack2(s0[1]), y23 += DT::unpack2(s1[1]);
                    }
                    auto p0 = __builtin_amdgcn_permlane16_swap(DT::pack2(x01), DT::pack2(y01), false, false);
                    auto p1 = __builtin_amdgcn_permlane16_swap(DT::pack2(x23), DT::pack2(y23), false, false);
                    const u32x4 v = u32x4{p0[0], p1[0], p0[1], p1[1]};
                    if constexpr (HAS_RES) {
                        // no store before the last residual load has been consumed (vmcnt retires in order, stores included)
                        outv[t] = v;
                        if (RD + t < NU) load_res(RD + t);
                    } else {
                        // unconditional buffer store, out-of-image lanes carry an offset past num_records (no branch per store, no
                        // 64-bit address per lane)
                        __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, (ok ? vo : 0x80000000u) + (unsigned)((cbase + m * 16) >> 4) * plane_bytes, 0, 0);
                    }
                }
            }
            if constexpr (HAS_RES) {
#pragma unroll
                for (int np = 0; np < NT / 2; ++np) {
                    bool ok;
                    unsigned vo = unit_off(np, false, ok);
                    if (!ok) vo = 0x80000000u;  // past num_records: dropped by the hardware
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        __builtin_amdgcn_raw_buffer_store_b128(outv[np * MT + m], orsrc, vo + (unsigned)((cbase + m * 16) >> 4) * plane_bytes, 0, 0);
                }
            }
        }
        if (tid == 0) {
            mbox[(seq_c + D) & 7] = fetched;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        AY_CLK(if (clk) { tk_epi += wall_clock64() - tk0; ++tk_items; })
        if (!has_next) break;
        item = next_item;
        ++seq_c;
        par = (par + 1) & 3;
    }
    leave();
    AY_CLK(if (clk && lane == 0) {
        const unsigned long long tot = wall_clock64() - tk_begin;
        atomicAdd(&g_phase_ticks_m16[0], tk_stage); atomicAdd(&g_phase_ticks_m16[1], tk_epi); atomicAdd(&g_phase_ticks_m16[2], tk_items);
        atomicAdd(&g_phase_ticks_m16[3], 1ull); atomicAdd(&g_phase_ticks_m16[4], tot); atomicAdd(&g_phase_ticks_m16[5], tk_wait);
        atomicMax(&g_phase_ticks_m16[6], tot); atomicAdd(&g_phase_ticks_m16[7], tk_str);
    })
}

}  // namespace ay

// 3x3 stride-1, cout_pad a multiple of 128, cin a multiple of 32, output rows >= 16: the 16x16x32-MFMA ring kernel.
// Same arguments and results as ay_conv_fwd_bf16 for those shapes (which calls it); returns AY_ERR_ARG for any other shape.
namespace ay {
template <typename DT>
static int conv3x3_m16_fwd(const ay_conv_desc* d, const void* src, const void* w_packed, const float* scale, const float* shift,
                           const void* residual, void* out, ay_stream_t stream) {
    AY_CHECK_ARG(d && src && w_packed && scale && shift && out, "ay_conv3x3_m16_fwd_bf16: null argument");
    AY_CHECK_ARG(d->ksize == 3 && d->stride == 1 && !d->out_f32 && d->cin % 32 == 0 && d->cout_pad % 128 == 0 && d->cout_pad >= d->cout,
                 "ay_conv3x3_m16_fwd_bf16: shape %dx%d k%d s%d", d->cin, d->cout_pad, d->ksize, d->stride);
    AY_CHECK_ARG(d->hout == d->hin && d->wout == d->win, "ay_conv3x3_m16_fwd_bf16: output size mismatch");
    AY_CHECK_ARG((long long)d->hout * d->wout * 2 * d->cout_pad < (1ll << 31),
                 "ay_conv3x3_m16_fwd_bf16: one image's output (%dx%dx%d) exceeds the 2 GiB a store descriptor addresses", d->hout, d->wout,
                 d->cout_pad);
    ConvArgs a;
    a.src = (const uint8_t*)src;
    a.w = (const uint8_t*)w_packed;
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
    a.tiles_x = (d->wout + 31) / 32;
    a.tiles_y = (d->hout + 15) / 16;
    a.n_cgroups = d->cout_pad / 128;
    a.leaky = d->leaky;
    a.dbg = 0;
#ifdef AY_PHASE_CLOCK
    static const int dbg = getenv("AY_DBG") ? atoi(getenv("AY_DBG")) : 0;
    a.dbg = dbg;
#endif
    a.stagger = 0;
    a.src1 = nullptr;
    a.c1 = 0;
    a.canvas_gx = 0;
    hipStream_t st = S(stream);
    a.deal = next_deal_set(st);
    const long long nblk = (long long)a.tiles_x * a.tiles_y * d->batch * a.n_cgroups;
    if (nblk <= 0 || nblk > 0x7fffffffLL) {
        set_error("conv grid out of range (%lld)", nblk);
        return AY_ERR_ARG;
    }
    const int per_xcd = (int)((nblk + 7) / 8);
    const int cu_slots = conv_num_cus() / 8;
    dim3 pgrid((unsigned)(8 * (per_xcd < cu_slots ? per_xcd : cu_slots)));
    if (residual)
        hipLaunchKernelGGL((conv3x3_m16_ring_kernel<true, DT>), pgrid, dim3(512), 0, st, a, (int)nblk);
    else
        hipLaunchKernelGGL((conv3x3_m16_ring_kernel<false, DT>), pgrid, dim3(512), 0, st, a, (int)nblk);
    AY_CHECK_LAUNCH("conv3x3_m16_ring_kernel");
#ifdef AY_PHASE_CLOCK
    if (a.dbg & 8) {  // timing experiments only: synchronous phase report per launch
        unsigned long long t[8] = {0};
        (void)hipStreamSynchronize(st);
        (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_phase_ticks_m16), sizeof(t));
        if (t[3])
            fprintf(stderr, "[ay phase m16] cin%d cout%d h%d res%d: items/wg %.1f  per item: stages %.2f us (of it waits for landed stages %.2f, straddle+barrier %.2f; %d stages), epilogue %.2f us; wg total %.1f us (slowest %.1f)\n",
                    d->cin, d->cout, d->hout, residual ? 1 : 0, (double)t[2] / t[3], t[0] * 0.01 / t[2], t[5] * 0.01 / t[2], t[7] * 0.01 / t[2],
                    d->cin / 16, t[1] * 0.01 / t[2], t[4] * 0.01 / t[3], t[6] * 0.01);
        unsigned long long z[8] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_ticks_m16), z, sizeof(z));
    }
#endif
    return AY_OK;
}
}  // namespace ay

extern "C" int ay_conv3x3_m16_fwd_bf16(const ay_conv_desc* d, const void* src, const void* w_packed, const float* scale, const float* shift,
                                       const void* residual, void* out, ay_stream_t stream) {
    return ay::conv3x3_m16_fwd<ay::Bf16>(d, src, w_packed, scale, shift, residual, out, stream);
}
extern "C" int ay_conv3x3_m16_fwd_f16(const ay_conv_desc* d, const void* src, const void* w_packed, const float* scale, const float* shift,
                                      const void* residual, void* out, ay_stream_t stream) {
    return ay::conv3x3_m16_fwd<ay::F16>(d, src, w_packed, scale, shift, residual, out, stream);
}
