// Fused Darknet-53 residual block (models.py:26-45 twice + the shortcut at :246-248):
//     mid = leaky(bn1(conv1x1(x)))   C -> C/2
//     out = leaky(bn2(conv3x3(mid))) + x   C/2 -> C
// in ONE persistent kernel, so `mid` (written and re-read 1.1x by the two-kernel path) never leaves the CU and the
// residual is re-read while its lines are still in L2.  HBM traffic per block: x once (+halo) and out once, 2.3 C-units
// instead of 4.1.  For the early, HBM-bound blocks where the 3x3 filters stream cheaply: measured at B=64 on the C=64
// block (512^2) 1.58 ms against 2.37 ms for the two launches; on the C=128 blocks (256^2) 1.15 against 1.04 ms (its
// stages are too short for their barriers), so the model fuses C=64 only (models.Darknet.fuse_block_channels).
//
// Per item (TH x 32 output pixels, all C channels), 8 waves:
//   A  1x1 on the (TH+2) x 34 halo tile: x and W1 stream through an NBUF-slot LDS-DMA ring in stages of NKA*16 input
//      channels; the C/2 x halo accumulator block is spread over the waves; epilogue A applies bn1 + leaky, rounds to bf16
//      ONCE, zeroes what lies outside the image (that is the 3x3's zero padding, not leaky(shift)) and writes the halo
//      tile of `mid` into LDS in the [chunk][half][pixel][8 ch] planes the 3x3 reads.
//   B  3x3 from those planes: only the filters stream through the ring (one 16-channel slab of 9 taps per stage); same
//      fragment maps, DMA spreading and epilogue (conv_epilogue: bn2 + leaky + residual + 16-byte stores) as
//      conv_bf16_ring_kernel.  Bit-identical to the two ay_conv_fwd_bf16 calls (same K order, same rounding points).
// The loader runs NBUF-1 stages ahead of the MFMAs across phase and item boundaries; every wave issues PW 1-KiB pieces
// per stage (dummy pieces pad the list) so "next stage landed" is one counted s_waitcnt.
#include "ay_conv_common.h"

namespace ay {

__device__ __attribute__((aligned(64))) uint32_t g_zero_page_rb[16];

struct ResBlockArgs {
    const uint8_t* x;    // [B][C/16][H][W][16] bf16: input and residual
    const uint8_t* w1;   // packed 1x1 filters [C/16][1][2][CM][8]
    const uint8_t* w2;   // packed 3x3 filters [CM/16][9][2][C][8]
    const float* scale1;
    const float* shift1;
    int leaky1;
    ConvArgs c2;         // the 3x3 as conv_epilogue sees it (out, residual = x, scale/shift, sizes)
};

template <int CM, int TH, int NKA, int NBUF, typename DT = Bf16>   // mid channels; tile rows; 16-channel chunks of x per phase-A stage; ring depth; storage type
__global__ void __launch_bounds__(512, 2) resblock_bf16_kernel(ResBlockArgs s, int n_items) {
    DT::enter();
    typedef typename DT::vec8 vec8;
    constexpr int C = 2 * CM;
    constexpr int TW = 32, IN_W = TW + 2, IN_H = TH + 2;
    constexpr int HP = IN_H * IN_W;                 // halo pixels (340 for 8 rows, 612 for 16)
    constexpr int NPB = (HP + 31) / 32;             // pixel blocks (11 / 20)
    constexpr int HPP = NPB * 32;                   // pixels per half-plane of `mid`
    constexpr int BN2 = C >= 128 ? 128 : 64;
    static_assert(C == BN2, "one output-channel pass");
    constexpr int SA = C / (16 * NKA);              // phase-A stages
    constexpr int SB = CM / 16;                     // phase-B stages
    constexpr int XP = (2 * HP + 63) / 64;          // pieces of one chunk of x: [half][HP px][16 B]
    constexpr int W1P = CM * 32 / 1024;             // pieces of one chunk of W1: [half][CM][16 B]
    static_assert(CM * 32 % 1024 == 0, "W1 chunk is whole pieces");
    constexpr int XSLAB = (XP + W1P) * 1024;        // one chunk in a phase-A slot: x pieces, then W1 pieces
    constexpr int A_PIECES = NKA * (XP + W1P);
    constexpr int B_PIECES = 9 * 2 * BN2 * 16 / 1024;
    constexpr int SLOT = (A_PIECES > B_PIECES ? A_PIECES : B_PIECES) * 1024;
    constexpr int PW = ((A_PIECES > B_PIECES ? A_PIECES : B_PIECES) + 7) / 8;
    constexpr int MID_SLAB = 2 * HPP * 16;
    constexpr int OFF_MID = NBUF * SLOT;
    constexpr int OFF_DUMMY = OFF_MID + (CM / 16) * MID_SLAB;
    constexpr int OFF_SS1 = OFF_DUMMY + 1024;       // [scale1 CM][shift1 CM] floats
    constexpr int OFF_SS2 = OFF_SS1 + 1024;         // [scale2 BN2 | pad to 128][shift2] floats (conv_epilogue's LDS format)
    constexpr int LDS_BYTES = OFF_SS2 + 1024;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    // phase A: waves = WMA (32-channel blocks of mid) x WNA (pixel-block residues)
    constexpr int WMA = CM / 32, WNA = 8 / WMA, NBA = (NPB + WNA - 1) / WNA;
    // phase B: as the ring kernel's 8x32 tile
    constexpr int WM = BN2 / 64, WN = 8 / WM, MT = 2, NT = (TH * TW) / (WN * 32);
    static_assert(MT * WM * 32 == BN2 && NT * WN * 32 == TH * TW, "tile split");
#ifndef AY_RESBLOCK_RES_LDS
#define AY_RESBLOCK_RES_LDS 1
#endif
    constexpr bool RES_FROM_LDS = AY_RESBLOCK_RES_LDS && MT * NT <= 4;   // 32 VGPRs of residual held from phase A to the epilogue

    __shared__ __attribute__((aligned(16))) uint8_t lds[LDS_BYTES];
    __builtin_amdgcn_s_setprio(2);  // above a co-resident merge-NMS wavefront (priority 0)
    const ConvArgs& a = s.c2;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, hh = lane >> 5;
    const int wma = wave % WMA, wna = wave / WMA;
    const int wm = wave % WM, wn = wave / WM;

    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int per_xcd = (n_items + 7) >> 3;
    const int first = xcd * per_xcd;
    const int last = min(first + per_xcd, n_items);
    int item = first + slot;
    if (item >= last) return;

    const int H = a.hout, W = a.wout;
    const size_t plane_bytes = (size_t)H * W * 32;
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const uint8_t* zero_page = reinterpret_cast<const uint8_t*>(g_zero_page_rb);
    const unsigned lds_base = lds_addr_of(lds);

    // per-channel affine of both layers: resident for the whole kernel
    {
        float* ss1 = reinterpret_cast<float*>(lds + OFF_SS1);
        float* ss2 = reinterpret_cast<float*>(lds + OFF_SS2);
        if (tid < CM) {
            ss1[tid] = s.scale1[tid];
            ss1[CM + tid] = s.shift1[tid];
        }
        if (tid < BN2) {
            ss2[tid] = a.scale[tid];
            ss2[128 + tid] = a.shift[tid];
        }
    }

    // ---- loader ---------------------------------------------------------------------------------------------------
    int xoff[PW];                    // per lane: byte offset of its 16 bytes inside a 16-channel plane of x, or -1 (zero page)
    const uint8_t* ld_xbase = nullptr;
    int ld_item = item, ld_ls = 0;   // stage within the item: 0..SA-1 phase A, SA..SA+SB-1 phase B
    bool ld_done = false;
    auto setup_loader = [&](int it) __attribute__((always_inline)) {
        const int b = it / tiles_per_img;
        const int y0 = ((it / a.tiles_x) % a.tiles_y) * TH, x0 = (it % a.tiles_x) * TW;
        ld_xbase = s.x + (size_t)b * (C / 16) * plane_bytes;
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int q = i * 8 + wave;
            int off = -1;
            if (q < A_PIECES) {
                const int r = q % (XP + W1P);
                if (r < XP) {
                    const int u = r * 64 + lane;
                    const int h = u / HP, P = u % HP;
                    const int iy = y0 - 1 + P / IN_W, ix = x0 - 1 + P % IN_W;
                    if (h < 2 && iy >= 0 && iy < H && ix >= 0 && ix < W) off = (iy * W + ix) * 32 + h * 16;
                }
            }
            xoff[i] = off;
        }
    };
    auto issue_piece = [&](int i, int buf) __attribute__((always_inline)) {
        const int q = i * 8 + wave;  // wave-uniform
        const uint8_t* g = zero_page + (lane & 3) * 16;
        int dst = OFF_DUMMY;
        if (ld_ls < SA) {
            if (q < A_PIECES) {
                const int kk = q / (XP + W1P), r = q % (XP + W1P);
                const int chunk = ld_ls * NKA + kk;
                if (r < XP) {
                    if (xoff[i] >= 0) g = ld_xbase + (size_t)chunk * plane_bytes + xoff[i];
                    dst = buf * SLOT + kk * XSLAB + r * 1024;
                } else {
                    g = s.w1 + (size_t)chunk * (CM * 32) + (r - XP) * 1024 + lane * 16;
                    dst = buf * SLOT + kk * XSLAB + r * 1024;
                }
            }
        } else if (q < B_PIECES) {
            const int kk = ld_ls - SA;
            const int u = q * 64 + lane;  // unit of the slab [tap][half][BN2]
            g = s.w2 + (size_t)kk * (9 * 2 * C * 16) + ((u / BN2) * C + (u % BN2)) * 16;
            dst = buf * SLOT + q * 1024;
        }
        dma16(g, lds_base + __builtin_amdgcn_readfirstlane(dst));
    };
    auto advance_loader = [&]() __attribute__((always_inline)) {
        if (++ld_ls == SA + SB) {
            ld_ls = 0;
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
    // "the next stage has landed": everything but the pieces of the NBUF-2 stages issued after it (a stage is only issued
    // if all earlier ones were, so `issued` for this stage means the whole window is in flight)
    auto stage_end = [&](bool issued, bool more) __attribute__((always_inline)) {
        if (!more) return;
        if (issued)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * PW) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // phase-B fragment addresses (same maps as conv_bf16_ring_kernel, 8x32 tile, stride 1)
    int pb[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int p = (wn * NT + n) * 32 + c;
        pb[n] = (hh * HPP + (p / TW) * IN_W + (p % TW)) * 16;
    }
    // residual unit of output pixel p in a chunk's x slab [half][HP px][16 B]: halo pixel (ty + 1, tx + 1)
    int rp[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int p = (wn * NT + n) * 32 + c;
        rp[n] = (hh * HP + (p / TW + 1) * IN_W + (p % TW + 1)) * 16;
    }
    const int wa2 = (hh * BN2 + wm * MT * 32 + c) * 16;
    const int wa1 = XP * 1024 + (hh * CM + wma * 32 + c) * 16;

    // ---- prologue: NBUF-1 stages in flight, stage 0 landed -----------------------------------------------------------
    static_assert(SA + SB >= NBUF - 1, "an item has at least NBUF-1 stages");
    setup_loader(item);
#pragma unroll
    for (int k = 0; k < NBUF - 1; ++k) issue_stage(k);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * PW) : "memory");
    __syncthreads();  // also publishes the scale/shift tables

    int cur = 0;
    while (true) {
        const int b = item / tiles_per_img;
        const int y0 = ((item / a.tiles_x) % a.tiles_y) * TH, x0 = (item % a.tiles_x) * TW;
        const int next_item = item + slots;
        const bool has_next = next_item < last;

        ResRegs<MT, NT> rr;
        // ================= phase A: mid = leaky(bn1(W1 . x)) on the halo tile =================
        f32x16 accA[NBA];
#pragma unroll
        for (int j = 0; j < NBA; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) accA[j][r] = 0.f;
        for (int sa = 0; sa < SA; ++sa) {
            const bool issued = !ld_done;
            int slot_ld = cur + (NBUF - 1);
            if (slot_ld >= NBUF) slot_ld -= NBUF;
            const uint8_t* L = lds + cur * SLOT;
            // the shortcut operand of the epilogue is x itself: this stage's 16-channel chunk(s) of the halo tile lie in the ring slot
            // right now, so the lanes pick their residual units (store layout: 16 bytes = channels 8hh.. of one pixel) out of LDS
            // instead of re-reading them from L2 / HBM 20 us later, when 32 CUs x 150 KB per item have pushed them out of the 4-MB
            // slice (the kernel fetched its input 1.91x: profiles/r02_hbm_traffic_v7.txt)
            if constexpr (RES_FROM_LDS) {
#pragma unroll
                for (int kk = 0; kk < NKA; ++kk) {
                    const int chunk = sa * NKA + kk;           // 16-channel plane of x
                    if ((chunk >> 2) == wm) {                  // this wave's 64 output channels (wave-uniform)
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            const uint4 v = *reinterpret_cast<const uint4*>(L + kk * XSLAB + rp[n]);
                            // rr.r[m][n][qp], m = (chunk >> 1) & 1, qp = chunk & 1: written through constant indices
                            if (((chunk >> 1) & 1) == 0) {
                                if ((chunk & 1) == 0) rr.r[0][n][0] = v; else rr.r[0][n][1] = v;
                            } else {
                                if ((chunk & 1) == 0) rr.r[1][n][0] = v; else rr.r[1][n][1] = v;
                            }
                        }
                    }
                }
            }
            vec8 fa[NKA], fb[NKA][NBA];
#pragma unroll
            for (int kk = 0; kk < NKA; ++kk) {
                fa[kk] = *reinterpret_cast<const vec8*>(L + kk * XSLAB + wa1);
#pragma unroll
                for (int j = 0; j < NBA; ++j)
                    fb[kk][j] = *reinterpret_cast<const vec8*>(L + kk * XSLAB + (hh * HP + (wna + j * WNA) * 32 + c) * 16);
            }
#pragma unroll
            for (int kk = 0; kk < NKA; ++kk) {
#pragma unroll
                for (int j = 0; j < NBA; ++j) accA[j] = DT::mfma32(fa[kk], fb[kk][j], accA[j]);
                if (issued) {
#pragma unroll
                    for (int i = 0; i < PW; ++i)
                        if (i >= kk * PW / NKA && i < (kk + 1) * PW / NKA) issue_piece(i, slot_ld);
                }
            }
            if (issued) advance_loader();
            stage_end(issued, true);
            if (++cur == NBUF) cur = 0;
        }
        // epilogue A: rows = mid channels wma*32 + (reg&3) + 8*(reg>>2) + 4*hh, column = halo pixel
        {
            const float* ss1 = reinterpret_cast<const float*>(lds + OFF_SS1);
#pragma unroll
            for (int j = 0; j < NBA; ++j) {
                const int nb = wna + j * WNA;
                if (nb < NPB) {  // wave-uniform
                    const int P = nb * 32 + c;
                    const int gy = y0 - 1 + P / IN_W, gx = x0 - 1 + P % IN_W;
                    const bool real = P < HP && gy >= 0 && gy < H && gx >= 0 && gx < W;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 scv = *reinterpret_cast<const float4*>(ss1 + wma * 32 + 8 * q + 4 * hh);
                        const float4 shv = *reinterpret_cast<const float4*>(ss1 + CM + wma * 32 + 8 * q + 4 * hh);
                        const float sc[4] = {scv.x, scv.y, scv.z, scv.w}, sh[4] = {shv.x, shv.y, shv.z, shv.w};
                        float o[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float t = accA[j][4 * q + e] * sc[e] + sh[e];
                            if (s.leaky1) t = t > 0.f ? t : 0.1f * t;
                            o[e] = real ? t : 0.f;
                        }
                        // channel wma*32 + 8q + 4hh + e -> chunk wma*2 + (q>>1), half q&1, element 4hh+e
                        uint8_t* dst = lds + OFF_MID + (wma * 2 + (q >> 1)) * MID_SLAB + ((q & 1) * HPP + P) * 16 + hh * 8;
                        *reinterpret_cast<uint2*>(dst) = make_uint2(pack2_scalar<DT>(o[0], o[1]), pack2_scalar<DT>(o[2], o[3]));
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // `mid` complete before any wave's 3x3 reads it
        asm volatile("" ::: "memory");

        // ================= phase B: out = leaky(bn2(W2 * mid)) + x =================
        f32x16 acc[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
        for (int kb = 0; kb < SB; ++kb) {
            const bool last_stage = (kb + 1 == SB);
            const bool issued = !ld_done;
            int slot_ld = cur + (NBUF - 1);
            if (slot_ld >= NBUF) slot_ld -= NBUF;
            const uint8_t* L = lds + cur * SLOT;
            const uint8_t* M = lds + OFF_MID + kb * MID_SLAB;
            vec8 af[2][MT], bfr[2][NT];
            auto load_frags = [&](int tap, vec8 (&fa)[MT], vec8 (&fb)[NT]) __attribute__((always_inline)) {
                const int kh = tap / 3, kw = tap % 3;
#pragma unroll
                for (int m = 0; m < MT; ++m) fa[m] = *reinterpret_cast<const vec8*>(L + wa2 + (tap * 2 * BN2 + m * 32) * 16);
#pragma unroll
                for (int n = 0; n < NT; ++n) fb[n] = *reinterpret_cast<const vec8*>(M + pb[n] + (kh * IN_W + kw) * 16);
            };
            load_frags(0, af[0], bfr[0]);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                if (t + 1 < 9) load_frags(t + 1, af[(t + 1) & 1], bfr[(t + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
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
                    for (int i = 0; i < PW; ++i)
                        if (i >= t * PW / 9 && i < (t + 1) * PW / 9) issue_piece(i, slot_ld);
                }
            }
            if (issued) advance_loader();
            stage_end(issued, !(last_stage && !has_next));
            if (++cur == NBUF) cur = 0;
        }
        conv_epilogue<BN2, MT, NT, TW, false, true, !RES_FROM_LDS, 2, false, false, DT>(a, acc, rr, b, 0, wm, wn, c, hh, y0, x0,
                                                                         reinterpret_cast<const float*>(lds + OFF_SS2));
        if (!has_next) break;
        item = next_item;
    }
}

template <int CM, int TH, int NKA, int NBUF, typename DT>
static int launch_resblock(const ResBlockArgs& s, int n_items, hipStream_t st) {
    const int per_xcd = (n_items + 7) / 8;
    const int cu_slots = conv_num_cus() / 8;
    dim3 grid((unsigned)(8 * (per_xcd < cu_slots ? per_xcd : cu_slots)));
    hipLaunchKernelGGL((resblock_bf16_kernel<CM, TH, NKA, NBUF, DT>), grid, dim3(512), 0, st, s, n_items);
    AY_CHECK_LAUNCH("resblock_bf16_kernel");
    return AY_OK;
}

}  // namespace ay

extern "C" int ay_resblock_supported(int channels) { return channels == 64 || channels == 128; }

namespace ay {
template <typename DT>
static int resblock_fwd(const void* x, const void* w1_packed, const float* scale1, const float* shift1, int leaky1,
                        const void* w2_packed, const float* scale2, const float* shift2, int leaky2, void* out, int batch,
                        int channels, int h, int w, ay_stream_t stream) {
    AY_CHECK_ARG(x && w1_packed && scale1 && shift1 && w2_packed && scale2 && shift2 && out, "ay_resblock_fwd_bf16: null");
    AY_CHECK_ARG(ay_resblock_supported(channels), "ay_resblock_fwd_bf16: %d channels unsupported (64 or 128)", channels);
    AY_CHECK_ARG(batch > 0 && h > 0 && w > 0 && x != out, "ay_resblock_fwd_bf16: bad shape / in-place");
    AY_CHECK_ARG((long long)h * w * 2 * channels < (1ll << 31), "ay_resblock_fwd_bf16: one image exceeds the 2 GiB a buffer descriptor addresses");
    ResBlockArgs s;
    s.x = (const uint8_t*)x;
    s.w1 = (const uint8_t*)w1_packed;
    s.w2 = (const uint8_t*)w2_packed;
    s.scale1 = scale1;
    s.shift1 = shift1;
    s.leaky1 = leaky1;
    ConvArgs& a = s.c2;
    a.src = nullptr;
    a.w = (const uint8_t*)w2_packed;
    a.scale = scale2;
    a.shift = shift2;
    a.residual = (const uint8_t*)x;
    a.out = (uint8_t*)out;
    a.batch = batch;
    a.cin = channels / 2;
    a.cout_pad = channels;
    a.hin = a.hout = h;
    a.win = a.wout = w;
    // C = 64: 16x32 tile, one x chunk per stage, 4-deep ring (HBM-latency bound: 3 stages in flight); C = 128: 8x32, 3-deep
    const int th = channels == 64 ? 16 : 8;
    a.tiles_x = (w + 31) / 32;
    a.tiles_y = (h + th - 1) / th;
    a.n_cgroups = 1;
    a.leaky = leaky2;
    a.dbg = 0;
    a.stagger = 0;
    a.deal = nullptr;
    a.canvas_gx = 0;
    a.src1 = nullptr;
    a.c1 = 0;
    const long long n_items = (long long)a.tiles_x * a.tiles_y * batch;
    AY_CHECK_ARG(n_items > 0 && n_items < 0x7fffffffLL, "ay_resblock_fwd_bf16: grid");
    return channels == 128 ? launch_resblock<64, 8, 2, 3, DT>(s, (int)n_items, S(stream)) : launch_resblock<32, 16, 1, 4, DT>(s, (int)n_items, S(stream));
}
}  // namespace ay

extern "C" int ay_resblock_fwd_bf16(const void* x, const void* w1_packed, const float* scale1, const float* shift1, int leaky1,
                                    const void* w2_packed, const float* scale2, const float* shift2, int leaky2, void* out, int batch,
                                    int channels, int h, int w, ay_stream_t stream) {
    return ay::resblock_fwd<ay::Bf16>(x, w1_packed, scale1, shift1, leaky1, w2_packed, scale2, shift2, leaky2, out, batch, channels, h, w, stream);
}
extern "C" int ay_resblock_fwd_f16(const void* x, const void* w1_packed, const float* scale1, const float* shift1, int leaky1,
                                   const void* w2_packed, const float* scale2, const float* shift2, int leaky2, void* out, int batch,
                                   int channels, int h, int w, ay_stream_t stream) {
    return ay::resblock_fwd<ay::F16>(x, w1_packed, scale1, shift1, leaky1, w2_packed, scale2, shift2, leaky2, out, batch, channels, h, w, stream);
}
