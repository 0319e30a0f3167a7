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
};

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

template <int BN, int MT, int NT, int TW, bool HAS_RES>
__device__ __forceinline__ void residual_prefetch(const ConvArgs& a, ResRegs<MT, NT>& rr, int b, int cg, int wm, int wn, int c,
                                                  int hh, int y0, int x0) {
    if constexpr (HAS_RES) {
        const int CP = a.cout_pad;
        const size_t out_plane_px = (size_t)a.hout * a.wout;
        const int cbase = cg * BN + wm * MT * 32;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int p = (wn * NT + n) * 32 + c;
            // clamped pixel: an always-valid address, so the loads are unconditional (a load under `if (ok)`, or an address
            // select the compiler turns into a branch, is waited for on the spot and serialises the residual stream);
            // pixels outside the image are never stored
            const int oy = min(y0 + p / TW, a.hout - 1), ox = min(x0 + p % TW, a.wout - 1);
            const size_t pix = (size_t)oy * a.wout + ox;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int qp = 0; qp < 2; ++qp) {
                    const int ch0 = cbase + m * 32 + qp * 16;
                    const size_t plane = (size_t)b * (CP / 16) + (ch0 >> 4);
                    rr.r[m][n][qp] = *reinterpret_cast<const uint4*>(a.residual + (plane * out_plane_px + pix) * 32 + hh * 16);
                }
        }
    }
}

// RES_INLINE: the residual is loaded here, one 32-pixel block ahead of its use (large wave tiles cannot hold all of it).
// SS_MODE 1|2 (ss_lds != nullptr; 0 = from global memory): per-channel scale/shift of this workgroup's BN channels staged in LDS as [scale BN | pad to 128][shift] (shift at float offset max(BN, 128)),
// so the epilogue issues no vector-memory loads that would have to wait behind its own stores.
template <int BN, int MT, int NT, int TW, bool OUT_F32, bool HAS_RES, bool RES_INLINE = false, int SS_MODE = 0>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[MT][NT], const ResRegs<MT, NT>& rr, int b, int cg,
                                              int wm, int wn, int c, int hh, int y0, int x0, const float* ss_lds = nullptr) {
    const int CP = a.cout_pad;
    const size_t out_plane_px = (size_t)a.hout * a.wout;
    const int cbase = cg * BN + wm * MT * 32;
    const int lbase = wm * MT * 32;  // channel index inside the workgroup's BN channels
    constexpr int SHO = BN > 128 ? BN : 128;  // float offset of the shifts in the LDS scale/shift image

    // residual of (n, m) 32x32 blocks, loaded RD blocks ahead of their use: a load issued only one block (~150 cycles)
    // ahead exposes nearly the whole memory latency on every block; the MFMA fragment registers are dead here, so the
    // deeper ring costs no extra registers
    constexpr int RD = (MT * NT >= 4) ? 2 : MT * NT;
    uint4 rres[RD][2];
    auto load_res = [&](int t, uint4 (&r)[2]) __attribute__((always_inline)) {
        const int n = t / MT, m = t % MT;
        const int p = (wn * NT + n) * 32 + c;
        const int oy = min(y0 + p / TW, a.hout - 1), ox = min(x0 + p % TW, a.wout - 1);  // clamped: see residual_prefetch
        const size_t pix = (size_t)oy * a.wout + ox;
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
            const size_t plane = (size_t)b * (CP / 16) + ((cbase + m * 32 + qp * 16) >> 4);
            r[qp] = *reinterpret_cast<const uint4*>(a.residual + (plane * out_plane_px + pix) * 32 + hh * 16);
        }
    };
    if constexpr (HAS_RES && RES_INLINE) {
#pragma unroll
        for (int t = 0; t < RD; ++t) load_res(t, rres[t]);
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int p = (wn * NT + n) * 32 + c;
        const int oy = y0 + p / TW, ox = x0 + p % TW;
        const bool ok = (oy < a.hout) && (ox < a.wout);
        const size_t pix = (size_t)oy * a.wout + ox;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            uint4 rcur[2];
            if constexpr (HAS_RES && RES_INLINE) {
                rcur[0] = rres[(n * MT + m) % RD][0];
                rcur[1] = rres[(n * MT + m) % RD][1];
                if (n * MT + m + RD < NT * MT) load_res(n * MT + m + RD, rres[(n * MT + m) % RD]);
            }
#pragma unroll
            for (int qp = 0; qp < 2; ++qp) {  // quad pair (2qp, 2qp+1) -> 16-channel plane
                const int ch0 = cbase + m * 32 + qp * 16;  // first channel of the plane
                float v[4], w[4];
                {
                    float4 s0, t0, s1, t1;
                    if constexpr (SS_MODE == 2) {
                        // explicit LDS address space: through a generic pointer the compiler may emit flat loads, which
                        // count on vmcnt too and make it wait for every LDS-DMA in flight (seen in the fused stem)
                        typedef __attribute__((address_space(3))) const f32x4 lds_f4;
                        lds_f4* sl = (lds_f4*)(__attribute__((address_space(3))) const float*)ss_lds;
                        const int l0 = lbase + m * 32 + qp * 16 + 4 * hh;  // multiple of 4 floats
                        const f32x4 a0 = sl[l0 >> 2], b0 = sl[(SHO + l0) >> 2], a1 = sl[(l0 + 8) >> 2], b1 = sl[(SHO + l0 + 8) >> 2];
                        s0 = make_float4(a0[0], a0[1], a0[2], a0[3]);
                        t0 = make_float4(b0[0], b0[1], b0[2], b0[3]);
                        s1 = make_float4(a1[0], a1[1], a1[2], a1[3]);
                        t1 = make_float4(b1[0], b1[1], b1[2], b1[3]);
                    } else if (ss_lds) {  // SS_MODE 1: LDS through the generic pointer, run-time test kept (resolves to ds_read in the
                                          // ring kernels; the code shape matters there: they sit at 256 VGPRs without a spill)
                        const int l0 = lbase + m * 32 + qp * 16 + 4 * hh;
                        s0 = *reinterpret_cast<const float4*>(ss_lds + l0);
                        t0 = *reinterpret_cast<const float4*>(ss_lds + SHO + l0);
                        s1 = *reinterpret_cast<const float4*>(ss_lds + l0 + 8);
                        t1 = *reinterpret_cast<const float4*>(ss_lds + SHO + l0 + 8);
                    } else {
                        s0 = *reinterpret_cast<const float4*>(a.scale + ch0 + 4 * hh);
                        t0 = *reinterpret_cast<const float4*>(a.shift + ch0 + 4 * hh);
                        s1 = *reinterpret_cast<const float4*>(a.scale + ch0 + 8 + 4 * hh);
                        t1 = *reinterpret_cast<const float4*>(a.shift + ch0 + 8 + 4 * hh);
                    }
                    const float ss0[4] = {s0.x, s0.y, s0.z, s0.w}, tt0[4] = {t0.x, t0.y, t0.z, t0.w};
                    const float ss1[4] = {s1.x, s1.y, s1.z, s1.w}, tt1[4] = {t1.x, t1.y, t1.z, t1.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float x0v = acc[m][n][(2 * qp) * 4 + j] * ss0[j] + tt0[j];
                        float x1v = acc[m][n][(2 * qp + 1) * 4 + j] * ss1[j] + tt1[j];
                        if (a.leaky) {
                            x0v = x0v > 0.f ? x0v : 0.1f * x0v;
                            x1v = x1v > 0.f ? x1v : 0.1f * x1v;
                        }
                        v[j] = x0v;
                        w[j] = x1v;
                    }
                }
                const size_t plane = (size_t)b * (CP / 16) + (ch0 >> 4);
                if constexpr (OUT_F32) {
                    // [plane][pixel][16 f32]: quad 2qp -> elems 4hh.., quad 2qp+1 -> elems 8+4hh..
                    if (ok) {
                        float* o = reinterpret_cast<float*>(a.out) + (plane * out_plane_px + pix) * 16;
                        *reinterpret_cast<float4*>(o + 4 * hh) = make_float4(v[0], v[1], v[2], v[3]);
                        *reinterpret_cast<float4*>(o + 8 + 4 * hh) = make_float4(w[0], w[1], w[2], w[3]);
                    }
                } else {
                    const size_t ob = (plane * out_plane_px + pix) * 32 + hh * 16;
                    if constexpr (HAS_RES) {
                        // residual: store layout -> accumulator layout, added in fp32 before the single bf16 rounding
                        uint4 rv;
                        if constexpr (RES_INLINE)
                            rv = rcur[qp];
                        else
                            rv = rr.r[m][n][qp];
                        auto sx = __builtin_amdgcn_permlane32_swap(rv.x, rv.z, false, false);
                        auto sy = __builtin_amdgcn_permlane32_swap(rv.y, rv.w, false, false);
                        const uint2 r0v = make_uint2(sx[0], sy[0]), r1v = make_uint2(sx[1], sy[1]);
                        v[0] += bf2f((uint16_t)(r0v.x & 0xffffu));
                        v[1] += bf2f((uint16_t)(r0v.x >> 16));
                        v[2] += bf2f((uint16_t)(r0v.y & 0xffffu));
                        v[3] += bf2f((uint16_t)(r0v.y >> 16));
                        w[0] += bf2f((uint16_t)(r1v.x & 0xffffu));
                        w[1] += bf2f((uint16_t)(r1v.x >> 16));
                        w[2] += bf2f((uint16_t)(r1v.y & 0xffffu));
                        w[3] += bf2f((uint16_t)(r1v.y >> 16));
                    }
                    {
                        unsigned ax = pack2bf(v[0], v[1]), ay_ = pack2bf(v[2], v[3]);
                        unsigned bx = pack2bf(w[0], w[1]), by = pack2bf(w[2], w[3]);
                        auto r0 = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
                        auto r1 = __builtin_amdgcn_permlane32_swap(ay_, by, false, false);
                        // plain stores: `nt` (streaming) stores were measured slower -- their completion, which the next stage's
                        // counted DMA wait sits behind, takes longer (first stage of the next item 4.0 -> 5.1 us)
                        if (ok && !AY_DBGBIT(a, 4)) *reinterpret_cast<uint4*>(a.out + ob) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
                    }
                }
            }
        }
    }
}

int conv_num_cus();

}  // namespace ay
