// Pieces shared by the bf16 MFMA convolution kernels: argument block, residual registers, fused epilogue.
#pragma once
#include "ay_common.h"

// Timing-experiment hooks (AY_DBG bits: 1 no staging, 2 no MFMA phase, 4 no output stores, 8 phase clock, 64 MFMA-only) exist
// only in a library built with -DAY_PHASE_CLOCK (AY_PHASE_CLOCK=1 python build.py).  In the product build AY_DBGBIT() is a
// compile-time 0: run-time branches inside the unrolled MFMA loops cost the register-saturated kernels 20-30 VGPRs of spills.
#ifdef AY_PHASE_CLOCK
#define AY_DBGBIT(a, bit) ((a).dbg & (bit))
#else
#define AY_DBGBIT(a, bit) 0
#endif

namespace ay {

struct ConvArgs {
    const uint8_t* src;
    const uint8_t* w;
    const float* scale;
    const float* shift;
    const uint8_t* residual;
    uint8_t* out;
    int batch, cin, cout_pad, hin, win, hout, wout;
    int tiles_x, tiles_y, n_cgroups;
    int leaky;
    int dbg;  // timing experiments only (AY_DBG): 1 = no staging in the stage loop, 2 = no MFMA phase
    int stagger;  // ring kernel: start workgroup (slot & 3) after slot&3 x stagger x ~4 us, so that the CUs' epilogue
                  // (HBM) phases do not coincide
    const uint8_t* src1;  // ring kernel, CAT variant: the first c1 input channels come from this tensor at half resolution
    int c1;               // (nearest x2 upsample folded into the loader); `src` then holds channels c1..cin-1
    unsigned* deal;  // ring kernel: per-launch work counters (8 per-XCD item counters + 1 exit counter), nullptr = static dealing
    // Small images (13x13, 8x8 grids: the deep layers at the reference's default 416-px input) fill a third of a 8x32 pixel
    // tile.  canvas_gx > 0: the tile grid lies on a virtual canvas instead -- canvas_gx images side by side, the rest of the batch
    // below, one zero column / row between neighbours (their shared halo) -- and every tile pixel maps to (image, y, x) or to
    // nothing.  Stride-1 same-size convolutions only; per-lane image offsets are 32-bit (host-checked).
    int canvas_gx;
    // 2x2-window kernels (KS == 2: the data gradient of a 3x3 stride-2 convolution, one stride-1 sub-convolution per output pixel
    // parity class (py, px)): the item's channel-group index carries the class in its two low bits, the class's filter image
    // lies w_class_stride bytes after the previous one, and output pixel (oy, ox) of the class grid is pixel (2 oy + py, 2 ox + px)
    // of the [2 hout][2 wout] output planes
    unsigned w_class_stride;
};

// Fused YOLO decode of a detection head (ay_head_decode_fwd_*: models.py:127-172 applied in the epilogue of the head's 1x1
// convolution, conv_bf16_kernel<..., DECODE>): where the prediction rows go and what the decode needs.  Empty for every other launch.
struct DecodeArgs {
    float* pred;        // [B][n_total][K] float32
    int n_total;        // rows per image
    int row_offset;     // first row of this head
    int A, K;           // anchors of the head, 5 + classes
    float stride;       // img_dim / G as the reference computes it (models.py:119)
    float aw[6], ah[6]; // anchors in pixels
};

// canvas pixel -> image pixel; false: gutter / beyond the batch (reads as zero, is never stored)
__device__ __forceinline__ bool canvas_px(const ConvArgs& a, int cy, int cx, int& b, int& y, int& x) {
    const int H1 = a.hout + 1, W1 = a.wout + 1;
    const int iy = cy / H1, ix = cx / W1;
    y = cy - iy * H1;
    x = cx - ix * W1;
    b = iy * a.canvas_gx + ix;
    return cy >= 0 && cx >= 0 && y < a.hout && x < a.wout && ix < a.canvas_gx && b < a.batch;
}
// the INPUT canvas of a stride-S layer: cells S times as large (input pixel S*y - pad + tap of output pixel y stays inside its own
// cell or falls into the gutter, where it reads as the zero padding)
template <int S>
__device__ __forceinline__ bool canvas_px_in(const ConvArgs& a, int cy, int cx, int& b, int& y, int& x) {
    if constexpr (S == 1) {
        return canvas_px(a, cy, cx, b, y, x);
    } else {
        const int H1 = S * (a.hout + 1), W1 = S * (a.wout + 1);
        const int iy = cy / H1, ix = cx / W1;
        y = cy - iy * H1;
        x = cx - ix * W1;
        b = iy * a.canvas_gx + ix;
        return cy >= 0 && cx >= 0 && y < a.hin && x < a.win && ix < a.canvas_gx && b < a.batch;
    }
}

// ---- epilogue: affine + leaky (+ residual) -> direct stores -------------------------------------------
// C/D layout of 32x32: col (pixel) = lane&31, row (channel) = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
// A lane pair (c, c+32) holds 8 consecutive channels of pixel c per register-quad; one
// v_permlane32_swap per dword turns two quads into two full 16-byte stores (1 KiB per wave store).
// residual operand of one wave tile, in the STORE layout (lane (c, hh) holds the 16 bytes it will store to: channels
// 8hh..8hh+7 of its pixel, one coalesced 16-byte load); the same two permlane32 swaps that turn accumulator quads into
// store vectors are their own inverse and bring it back to the accumulator layout before the fp32 add.
template <int MT, int NT>
struct ResRegs {
    uint4 r[MT][NT][2];
};

// PAIR (with UP2): the workgroup's BN "channels" are the two column-parity classes (cls, cls + 1) of BN / 2 real channels each, side
// by side: block blk of 32 tile channels = class cls + (blk * 32) / (BN / 2), real channels cg * (BN / 2) + (blk * 32) % (BN / 2) ...  A
// wave then stores pixel (.., 2 ox) and pixel (.., 2 ox + 1) of a plane in consecutive instructions: whole lines leave the CU, where
// one class per workgroup leaves 32-byte segments at a 64-byte stride that reach HBM at 1.25 instead of 4.3-5.5 TB/s
// (scripts/micro/partial_line_stores.hip).
template <int BN, bool PAIR>
__device__ __forceinline__ void pair_block(int cg, int blk, int cls, int& ch_first, int& cls_out) {
    if constexpr (PAIR) {
        constexpr int HALF = BN / 2;
        ch_first = cg * HALF + (blk * 32) % HALF;
        cls_out = cls + (blk * 32) / HALF;
    } else {
        ch_first = cg * BN + blk * 32;
        cls_out = cls;
    }
}

template <int BN, int MT, int NT, int TW, bool HAS_RES, bool CANVAS = false, bool UP2 = false, bool PAIR = false>
__device__ __forceinline__ void residual_prefetch(const ConvArgs& a, ResRegs<MT, NT>& rr, int b, int cg, int wm, int wn, int c,
                                                  int hh, int y0, int x0, int cls = 0) {
    if constexpr (HAS_RES) {
        const int CP = a.cout_pad;
        const size_t out_plane_px = (size_t)a.hout * a.wout * (UP2 ? 4 : 1);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int p = (wn * NT + n) * 32 + c;
            // clamped pixel: an always-valid address, so the loads are unconditional (a load under `if (ok)`, or an address
            // select the compiler turns into a branch, is waited for on the spot and serialises the residual stream);
            // pixels outside the image are never stored
            int oy = min(y0 + p / TW, a.hout - 1), ox = min(x0 + p % TW, a.wout - 1), bb = b;
            if constexpr (CANVAS) {
                if (!canvas_px(a, y0 + p / TW, x0 + p % TW, bb, oy, ox)) bb = oy = ox = 0;  // any valid address
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                int chf, clm;
                pair_block<BN, PAIR>(cg, wm * MT + m, cls, chf, clm);
                const size_t pix = UP2 ? (size_t)(2 * oy + (clm >> 1)) * (2 * a.wout) + 2 * ox + (clm & 1) : (size_t)oy * a.wout + ox;
#pragma unroll
                for (int qp = 0; qp < 2; ++qp) {
                    const int ch0 = chf + qp * 16;
                    const size_t plane = (size_t)bb * (CP / 16) + (ch0 >> 4);
                    rr.r[m][n][qp] = *reinterpret_cast<const uint4*>(a.residual + (plane * out_plane_px + pix) * 32 + hh * 16);
                }
            }
        }
    }
}

// RES_INLINE: the residual is loaded here, RD 32-pixel blocks ahead of its use (large wave tiles cannot hold all of it).
// SS_MODE 1|2 (ss_lds != nullptr; 0 = from global memory): per-channel scale/shift of this workgroup's BN channels staged in
// LDS as [scale BN | pad to 128][shift] (shift at float offset max(BN, 128)), so the epilogue issues no vector-memory loads
// that would have to wait behind its own stores.
// Deferred stores (residual loaded here): see below; they go out as unconditional buffer stores -- out-of-image lanes carry an
// offset past num_records and the hardware drops them -- so that the compiler can count them (a store under `if (ok)` sits in
// its own basic block and forces conservative waits).  The plane offset rides in the VECTOR offset: with it in the scalar-offset
// field hipcc (following the ISA manual) puts no wait state between a 128-bit store and a VALU write of its data registers, and
// gfx950 then stored the overwritten first dword for the last lanes of each row of 16 (seen in the parity tests).
//
// Arithmetic: packed fp32 (v_pk_mul_f32 / v_pk_add_f32 carry two values per issue slot), LeakyReLU as max(x, slope * x) with
// slope 1 for linear layers, one v_cvt_pk_bf16_f32 per two outputs.  The epilogue is VALU-bound -- with scalar ops, compare +
// select and one convert per value it cost ~85 VALU instructions per 8 outputs, 4-5 us per 16x32 item with no MFMA running.
// The IEEE operations and their order are unchanged (no contraction), so are the results, bit for bit.
// CANVAS: the tile lies on the canvas of ConvArgs::canvas_gx (a compile-time switch: as run-time branches the mapping code
// cost the kernels that never use it SGPR spills -- the fused block went from 1.36 to 1.81 ms)
template <int BN, int MT, int NT, int TW, bool OUT_F32, bool HAS_RES, bool RES_INLINE = false, int SS_MODE = 0, bool CANVAS = false,
          bool UP2 = false, typename DT = Bf16, bool PAIR = false>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[MT][NT], const ResRegs<MT, NT>& rr, int b, int cg,
                                              int wm, int wn, int c, int hh, int y0, int x0, const float* ss_lds = nullptr, int cls = 0) {
    static_assert(!UP2 || (!CANVAS && !OUT_F32), "parity-class output: plain bf16 tiles");
    static_assert(!PAIR || UP2, "class pairs exist for the parity-class output");
    const int CP = a.cout_pad;
    const size_t out_plane_px = (size_t)a.hout * a.wout * (UP2 ? 4 : 1);
    // pixel index of output (oy, ox) inside a plane
    auto opix = [&](int oy, int ox, int cl) __attribute__((always_inline)) {
        return UP2 ? (unsigned)(2 * oy + (cl >> 1)) * (unsigned)(2 * a.wout) + 2 * ox + (cl & 1) : (unsigned)oy * a.wout + ox;
    };
    // first real channel and parity class of this wave's m-th block of 32 tile channels (PAIR: see pair_block)
    auto blk_of = [&](int m, int& chf, int& clm) __attribute__((always_inline)) { pair_block<BN, PAIR>(cg, wm * MT + m, cls, chf, clm); };
    const int lbase = wm * MT * 32;  // channel index inside the workgroup's BN channels
    constexpr int SHO = BN > 128 ? BN : 128;  // float offset of the shifts in the LDS scale/shift image
    const float slope = a.leaky ? 0.1f : 1.0f;
    // buffer descriptor over image b's output; offsets >= num_records are dropped, so the out-of-image sentinel 0x80000000
    // (+ a plane offset) needs an image below 2 GiB: host check
    const unsigned plane_bytes = (unsigned)out_plane_px * 32u;
    const unsigned img_bytes = plane_bytes * (unsigned)(CP / 16);
    // canvas mode: a lane's image varies, the descriptors span the whole tensor (below 2 GiB there, host-checked) and the image
    // offset rides in the vector offset
    const unsigned n_img = CANVAS ? (unsigned)a.batch : 1u;
    const int b0 = CANVAS ? 0 : __builtin_amdgcn_readfirstlane(b);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)b0 * img_bytes, 0, (int)(img_bytes * n_img), 0x00020000);

    // residual loaded here (RES_INLINE), RD blocks ahead of its use: vmcnt retires in order, STORES INCLUDED, so a residual load
    // issued behind an output store cannot be waited for before that store has completed.  No store is therefore issued until
    // the last residual load has been consumed: the packed bf16 outputs stay in registers (8 per 32x32 block, in the place of
    // the block's 16 accumulators) and go out together at the end (-1.2 % on the 16x32 residual kernel).  A look-ahead that
    // grows with the registers freed this way spilled (2 blocks more per finished block: 15 VGPRs; 256 is the budget), and
    // warming L2 with one 4-byte LDS-DMA per residual line a stage ahead made the kernel 3 % SLOWER: the epilogue phase is
    // bound by the HBM burst of residual + output of all CUs, not by the latency of a single load.
    constexpr int NB = MT * NT;
    constexpr int RD = (NB >= 4) ? 2 : NB;
    constexpr bool DEFER = HAS_RES && RES_INLINE && !OUT_F32;
    uint4 rres[NB][2];
    u32x4 outv[DEFER ? NB : 1][2];
    // residual through a buffer descriptor as well: a 32-bit pixel offset per lane + the plane in the scalar offset (no 64-bit
    // address arithmetic in vector registers; this epilogue runs at the register limit)
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.residual) + (size_t)b0 * img_bytes, 0,
                                                                           (int)(img_bytes * n_img), 0x00020000);
    auto load_res = [&](int t, uint4 (&r)[2]) __attribute__((always_inline)) {
        const int n = t / MT, m = t % MT;
        const int p = (wn * NT + n) * 32 + c;
        int oy = min(y0 + p / TW, a.hout - 1), ox = min(x0 + p % TW, a.wout - 1), bb = 0;  // clamped: see residual_prefetch
        if constexpr (CANVAS) {
            if (!canvas_px(a, y0 + p / TW, x0 + p % TW, bb, oy, ox)) bb = oy = ox = 0;
        }
        int chf, clm;
        blk_of(m, chf, clm);
        const unsigned pix_off = opix(oy, ox, clm) * 32u + hh * 16u + (unsigned)bb * img_bytes;
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
            const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)((chf + qp * 16) >> 4) * plane_bytes);
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, pix_off, so, 0);
            r[qp] = make_uint4(v[0], v[1], v[2], v[3]);
        }
    };
    if constexpr (HAS_RES && RES_INLINE) {
#pragma unroll
        for (int t = 0; t < RD; ++t) load_res(t, rres[t]);
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int p = (wn * NT + n) * 32 + c;
        int oy = y0 + p / TW, ox = x0 + p % TW, bn = b;  // bn: the image of this lane's pixel (canvas mode: per lane)
        bool ok = (oy < a.hout) && (ox < a.wout) && !AY_DBGBIT(a, 4);
        if constexpr (CANVAS) ok = canvas_px(a, y0 + p / TW, x0 + p % TW, bn, oy, ox) && !AY_DBGBIT(a, 4);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int t = n * MT + m;
            int chf, clm;
            blk_of(m, chf, clm);
            const size_t pix = opix(oy, ox, clm);
#pragma unroll
            for (int qp = 0; qp < 2; ++qp) {  // quad pair (2qp, 2qp+1) -> 16-channel plane
                const int ch0 = chf + qp * 16;  // first channel of the plane
                f32x2 v01, v23, w01, w23;                  // quad 2qp, quad 2qp+1
                {
                    f32x4 s0, t0, s1, t1;
                    if constexpr (SS_MODE == 2) {
                        // explicit LDS address space: through a generic pointer the compiler may emit flat loads, which
                        // count on vmcnt too and make it wait for every LDS-DMA in flight (seen in the fused stem)
                        typedef __attribute__((address_space(3))) const f32x4 lds_f4;
                        lds_f4* sl = (lds_f4*)(__attribute__((address_space(3))) const float*)ss_lds;
                        const int l0 = lbase + m * 32 + qp * 16 + 4 * hh;  // multiple of 4 floats
                        s0 = sl[l0 >> 2], t0 = sl[(SHO + l0) >> 2], s1 = sl[(l0 + 8) >> 2], t1 = sl[(SHO + l0 + 8) >> 2];
                    } else if (ss_lds) {  // SS_MODE 1: LDS through the generic pointer, run-time test kept (resolves to ds_read in the
                                          // ring kernels; the code shape matters there: they sit at 256 VGPRs without a spill)
                        const int l0 = lbase + m * 32 + qp * 16 + 4 * hh;
                        s0 = *reinterpret_cast<const f32x4*>(ss_lds + l0);
                        t0 = *reinterpret_cast<const f32x4*>(ss_lds + SHO + l0);
                        s1 = *reinterpret_cast<const f32x4*>(ss_lds + l0 + 8);
                        t1 = *reinterpret_cast<const f32x4*>(ss_lds + SHO + l0 + 8);
                    } else {
                        s0 = *reinterpret_cast<const f32x4*>(a.scale + ch0 + 4 * hh);
                        t0 = *reinterpret_cast<const f32x4*>(a.shift + ch0 + 4 * hh);
                        s1 = *reinterpret_cast<const f32x4*>(a.scale + ch0 + 8 + 4 * hh);
                        t1 = *reinterpret_cast<const f32x4*>(a.shift + ch0 + 8 + 4 * hh);
                    }
                    const f32x16& q = acc[m][n];
                    constexpr int Q0 = 0, Q1 = 4;  // register offsets of the two quads inside the pair
                    const int o = qp * 8;
                    v01 = f32x2{q[o + Q0 + 0], q[o + Q0 + 1]} * f32x2{s0[0], s0[1]} + f32x2{t0[0], t0[1]};
                    v23 = f32x2{q[o + Q0 + 2], q[o + Q0 + 3]} * f32x2{s0[2], s0[3]} + f32x2{t0[2], t0[3]};
                    w01 = f32x2{q[o + Q1 + 0], q[o + Q1 + 1]} * f32x2{s1[0], s1[1]} + f32x2{t1[0], t1[1]};
                    w23 = f32x2{q[o + Q1 + 2], q[o + Q1 + 3]} * f32x2{s1[2], s1[3]} + f32x2{t1[2], t1[3]};
                    v01 = leaky2(v01, slope), v23 = leaky2(v23, slope), w01 = leaky2(w01, slope), w23 = leaky2(w23, slope);
                }
                const size_t plane = (size_t)bn * (CP / 16) + (ch0 >> 4);
                if constexpr (OUT_F32) {
                    // [plane][pixel][16 f32]: quad 2qp -> elems 4hh.., quad 2qp+1 -> elems 8+4hh..
                    if (ok) {
                        float* o = reinterpret_cast<float*>(a.out) + (plane * out_plane_px + pix) * 16;
                        *reinterpret_cast<float4*>(o + 4 * hh) = make_float4(v01[0], v01[1], v23[0], v23[1]);
                        *reinterpret_cast<float4*>(o + 8 + 4 * hh) = make_float4(w01[0], w01[1], w23[0], w23[1]);
                    }
                } else {
                    if constexpr (HAS_RES) {
                        // residual: store layout -> accumulator layout, added in fp32 before the single bf16 rounding
                        uint4 rv;
                        if constexpr (RES_INLINE)
                            rv = rres[t][qp];
                        else
                            rv = rr.r[m][n][qp];
                        auto sx = __builtin_amdgcn_permlane32_swap(rv.x, rv.z, false, false);
                        auto sy = __builtin_amdgcn_permlane32_swap(rv.y, rv.w, false, false);
                        v01 += DT::unpack2(sx[0]), v23 += DT::unpack2(sy[0]);
                        w01 += DT::unpack2(sx[1]), w23 += DT::unpack2(sy[1]);
                    }
                    const unsigned ax = DT::pack2(v01), ay_ = DT::pack2(v23), bx = DT::pack2(w01), by = DT::pack2(w23);
                    auto r0 = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
                    auto r1 = __builtin_amdgcn_permlane32_swap(ay_, by, false, false);
                    // plain stores: `nt` (streaming) stores were measured slower -- their completion, which the next stage's
                    // counted DMA wait sits behind, takes longer (first stage of the next item 4.0 -> 5.1 us)
                    if constexpr (DEFER) {
                        outv[t][qp] = u32x4{r0[0], r1[0], r0[1], r1[1]};
                    } else {
                        if (ok) *reinterpret_cast<uint4*>(a.out + (plane * out_plane_px + pix) * 32 + hh * 16) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
                    }
                }
            }
            if constexpr (HAS_RES && RES_INLINE) {  // this block's residual registers are free: the block RD ahead
                if (RD + t < NB) load_res(RD + t, rres[RD + t]);
            }
        }
    }
    if constexpr (DEFER) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int p = (wn * NT + n) * 32 + c;
            int oy = y0 + p / TW, ox = x0 + p % TW, bn = 0;
            bool ok = (oy < a.hout) && (ox < a.wout) && !AY_DBGBIT(a, 4);
            if constexpr (CANVAS) ok = canvas_px(a, y0 + p / TW, x0 + p % TW, bn, oy, ox) && !AY_DBGBIT(a, 4);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                int chf, clm;
                blk_of(m, chf, clm);
                const unsigned vo = ok ? opix(oy, ox, clm) * 32u + hh * 16u + (unsigned)bn * img_bytes : 0x80000000u;
#pragma unroll
                for (int qp = 0; qp < 2; ++qp)
                    __builtin_amdgcn_raw_buffer_store_b128(outv[n * MT + m][qp], orsrc, vo + (unsigned)((chf + qp * 16) >> 4) * plane_bytes, 0, 0);
            }
        }
    }
}

// Epilogue of a detection head with the decode fused in: the linear 1x1 block's value (acc * scale + shift, as conv_epilogue forms
// it; scale = 1, shift = the bias) never goes to memory as a head tensor -- each lane turns the values it holds for its grid cell
// straight into the entries of the prediction row (yolo_decode_kernel's expressions, operation for operation: the reference's
// order, no contraction) and stores them.  A value is channel ch = anchor * K + k of the head; decoding it needs nothing but
// itself, the cell and the anchor.  K == 8 (three classes): a lane's four consecutive channels are k = 0..3 or 4..7 of ONE anchor
// -- one 16-byte store; other K: four scalar stores.
template <int BN, int MT, int NT, int TW>
__device__ __forceinline__ void head_decode_epilogue(const ConvArgs& a, const DecodeArgs& dd, f32x16 (&acc)[MT][NT], int b, int cg, int wm,
                                                     int wn, int c, int hh, int y0, int x0) {
    const int G = a.hout;
    const float stride = dd.stride;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int p = (wn * NT + n) * 32 + c;
        const int oy = y0 + p / TW, ox = x0 + p % TW;
        if (!(oy < a.hout && ox < a.wout)) continue;
        float* rows = dd.pred + ((size_t)b * dd.n_total + dd.row_offset + (size_t)oy * G + ox) * dd.K;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {   // accumulator quad qd: channels ch0 .. ch0 + 3 of this lane
                const int ch0 = cg * BN + (wm * MT + m) * 32 + qd * 8 + 4 * hh;
                if (ch0 >= dd.A * dd.K) continue;
                const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + ch0);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + ch0);
                float res[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ch = ch0 + j;
                    const int an = ch / dd.K, k = ch - an * dd.K;
                    const float v = acc[m][n][qd * 4 + j] * sc[j] + sh[j];
                    const bool wh = (k == 2) || (k == 3);
                    const float e = expf(wh ? v : -v);
                    const float sig = 1.0f / (1.0f + e);
                    const float anc = (k == 2 ? dd.aw[an < 6 ? an : 0] : dd.ah[an < 6 ? an : 0]) / stride;
                    res[j] = k == 0 ? (sig + (float)ox) * stride : k == 1 ? (sig + (float)oy) * stride : wh ? (e * anc) * stride : sig;
                }
                if (dd.K == 8) {   // channels ch0..ch0+3 = entries 4hh'..4hh'+3 of anchor ch0 / 8
                    const int an = ch0 >> 3;
                    *reinterpret_cast<float4*>(rows + (size_t)an * G * G * 8 + (ch0 & 7)) = make_float4(res[0], res[1], res[2], res[3]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int ch = ch0 + j;
                        const int an = ch / dd.K, k = ch - an * dd.K;
                        if (ch < dd.A * dd.K) rows[(size_t)an * G * G * dd.K + k] = res[j];
                    }
                }
            }
        }
    }
}

int conv_num_cus();
// counter set for the dynamic item dealing of the next ring-kernel launch on `st` (nullptr: static dealing); ay_conv_bf16.hip
unsigned* next_deal_set(hipStream_t st);

}  // namespace ay
